// Place-recognition retrieval on the device — the step right after the descriptor path in the database build
// (BASELINE configs[4]; reference eval/evaluate.py:60-88 and :168-184):
//     embed_dist = np.linalg.norm(map_embeddings - query_embedding, axis=1);  nn_ndx = np.argsort(embed_dist)[:k]
//     euclid_dist = norm(query_pos - map_positions[nn_ndx]);  tp[r][nn] += any(euclid_dist[:nn+1] <= r)
// knn_dist_kernel: one workgroup per query, one wave per database row at a time (a 256-float row is one
// float4 per lane), exact difference form (not the |a|^2+|b|^2-2ab expansion) like the reference.
// knn_select_kernel: k rounds of a workgroup-wide arg-min over the query's distance row (ties: lower index).
#include "common.h"
#include "kernels.h"

namespace egonn {

__global__ __launch_bounds__(256) void knn_dist_kernel(const float* __restrict__ query, const float* __restrict__ db,
                                                      int32_t m, int d, float* __restrict__ dist) {
  extern __shared__ float s_q[];
  const int q = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (int i = t; i < d; i += 256) s_q[i] = query[(int64_t)q * d + i];
  __syncthreads();
  for (int32_t r = w; r < m; r += 4) {
    const float* row = db + (int64_t)r * d;
    float s = 0.f;
    for (int i = lane; i < d; i += 64) {
      const float df = row[i] - s_q[i];
      s = fmaf(df, df, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) dist[(int64_t)q * m + r] = sqrtf(s);
  }
}

__global__ __launch_bounds__(256) void knn_select_kernel(float* __restrict__ dist, int32_t m, int k,
                                                        int32_t* __restrict__ out_idx, float* __restrict__ out_dist) {
  __shared__ float s_v[4];
  __shared__ int32_t s_i[4];
  const int q = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
  float* row = dist + (int64_t)q * m;
  for (int round = 0; round < k; ++round) {
    float bv = INFINITY;
    int32_t bi = 0x7fffffff;
    for (int32_t i = t; i < m; i += 256) {
      const float v = row[i];
      if (v < bv || (v == bv && i < bi)) {
        bv = v;
        bi = i;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int32_t oi = __shfl_xor(bi, o, 64);
      if (ov < bv || (ov == bv && oi < bi)) {
        bv = ov;
        bi = oi;
      }
    }
    if (lane == 0) {
      s_v[w] = bv;
      s_i[w] = bi;
    }
    __syncthreads();
    if (t == 0) {
      for (int j = 1; j < 4; ++j)
        if (s_v[j] < s_v[0] || (s_v[j] == s_v[0] && s_i[j] < s_i[0])) {
          s_v[0] = s_v[j];
          s_i[0] = s_i[j];
        }
      const bool ok = s_i[0] != 0x7fffffff && round < m;
      out_idx[(int64_t)q * k + round] = ok ? s_i[0] : -1;
      out_dist[(int64_t)q * k + round] = ok ? s_v[0] : INFINITY;
      if (ok) row[s_i[0]] = INFINITY;             // taken
    }
    __syncthreads();
  }
}

int knn_search(const float* query, int32_t nq, const float* db, int32_t m, int d, int k, int32_t* out_idx, float* out_dist,
               float* scratch, size_t scratch_floats, hipStream_t stream) {
  EGONN_REQUIRE(query && db && out_idx && out_dist && nq >= 0 && m >= 1 && d >= 1 && d <= 4096 && k >= 1, EGONN_ERR_INVALID,
                "knn: bad arguments (nq=%d m=%d d=%d k=%d)", nq, m, d, k);
  if (nq == 0) return EGONN_OK;
  EGONN_REQUIRE(scratch && scratch_floats >= (size_t)nq * m, EGONN_ERR_INVALID,
                "knn: scratch needs %lld floats", (long long)nq * m);
  hipLaunchKernelGGL(knn_dist_kernel, dim3((unsigned)nq), dim3(256), (size_t)d * 4, stream, query, db, m, d, scratch);
  hipLaunchKernelGGL(knn_select_kernel, dim3((unsigned)nq), dim3(256), 0, stream, scratch, m, k, out_idx, out_dist);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// tp[r][nn] = number of queries with a retrieved neighbour among the first nn+1 whose position lies within radius[r]
__global__ void recall_kernel(const int32_t* __restrict__ nn_idx, const float* __restrict__ qpos,
                              const float* __restrict__ mpos, int32_t nq, int k, int pd, const float* __restrict__ radius,
                              int nr, int32_t* __restrict__ tp) {
  const int32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  float best = INFINITY;
  for (int nn = 0; nn < k; ++nn) {
    const int32_t j = nn_idx[(int64_t)q * k + nn];
    if (j >= 0) {
      float s = 0.f;
      for (int c = 0; c < pd; ++c) {
        const float df = qpos[(int64_t)q * pd + c] - mpos[(int64_t)j * pd + c];
        s = fmaf(df, df, s);
      }
      best = fminf(best, sqrtf(s));
    }
    for (int r = 0; r < nr; ++r)
      if (best <= radius[r]) atomicAdd(&tp[r * k + nn], 1);
  }
}

int recall_counts(const int32_t* nn_idx, const float* qpos, const float* mpos, int32_t nq, int k, int pd,
                  const float* radius, int nr, int32_t* tp, hipStream_t stream) {
  EGONN_REQUIRE(nn_idx && qpos && mpos && radius && tp && k >= 1 && nr >= 1 && pd >= 1, EGONN_ERR_INVALID,
                "recall: bad arguments");
  HIP_CHECK(hipMemsetAsync(tp, 0, (size_t)nr * k * 4, stream));
  if (nq == 0) return EGONN_OK;
  hipLaunchKernelGGL(recall_kernel, dim3((unsigned)cdiv(nq, 128)), dim3(128), 0, stream, nn_idx, qpos, mpos, nq, k, pd,
                     radius, nr, tp);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

}  // namespace egonn
