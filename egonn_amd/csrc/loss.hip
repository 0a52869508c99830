// Batch-hard triplet loss with boolean positive / negative masks on a (gathered) embedding matrix — the consumer
// of the RCCL all-gather in the sharded training step (BASELINE.json configs[3]).
//
// Reference: models/loss.py:95-172 — HardTripletMinerWithMasks.mine (:114-129), get_max_per_row / get_min_per_row
// (:132-143), BatchHardTripletLossWithMasks.__call__ (:156-172).  The distance / loss / reducer classes it calls
// live in pytorch_metric_learning (>= 1.0, absent from the image): LpDistance(p=2, normalize_embeddings=False),
// TripletMarginLoss(margin, swap=True), AvgNonZeroReducer — restated per SURVEY.md Appendix A.9 (parity unpinned).
//
//   D[i][j]   = || e_i - e_j ||_2
//   p(i)      = argmax_j D[i][j] over positives (masked-out entries count as 0), n(i) = argmin_j over negatives
//               (masked-out = +inf); anchors without a positive or without a negative are dropped
//   l_i       = relu(D[i][p] - min(D[i][n], D[p][n]) + margin)          (swap=True)
//   loss      = mean of the l_i > 0  (0 if none)                         (AvgNonZeroReducer)
// Also returns dLoss/dE (deterministic gather formulation) so that each rank can back-propagate its own rows.
#include "common.h"
#include "kernels.h"

namespace egonn {

// D[i][j]; block = row i
__global__ __launch_bounds__(256) void pdist_kernel(const float* __restrict__ e, int n, int d, float* __restrict__ D,
                                                    float* __restrict__ norms) {
  extern __shared__ float ei[];
  const int i = blockIdx.x;
  for (int c = threadIdx.x; c < d; c += blockDim.x) ei[c] = e[(int64_t)i * d + c];
  __syncthreads();
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const float* ej = e + (int64_t)j * d;
    float s = 0.f;
    for (int c = 0; c < d; ++c) {
      const float t = ei[c] - ej[c];
      s = fmaf(t, t, s);
    }
    D[(int64_t)i * n + j] = sqrtf(s);
  }
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int c = 0; c < d; ++c) s = fmaf(ei[c], ei[c], s);
    norms[i] = sqrtf(s);
  }
}

// hardest positive / negative of row i (first index on ties, like a sequential scan)
__global__ __launch_bounds__(256) void mine_kernel(const float* __restrict__ D, const uint8_t* __restrict__ pos,
                                                   const uint8_t* __restrict__ neg, int n,
                                                   int32_t* __restrict__ trip, float* __restrict__ hp,
                                                   float* __restrict__ hn) {
  __shared__ float s_v[2][256];
  __shared__ int s_i[2][256];
  __shared__ int s_any[2];
  const int i = blockIdx.x, t = threadIdx.x;
  if (t < 2) s_any[t] = 0;
  __syncthreads();
  float bp = -1.f, bn = INFINITY;
  int ip = n, in_ = n;
  bool ap = false, an = false;
  for (int j = t; j < n; j += 256) {
    const float dv = D[(int64_t)i * n + j];
    const bool mp = pos[(int64_t)i * n + j] != 0, mn = neg[(int64_t)i * n + j] != 0;
    const float vp = mp ? dv : 0.f;                 // mat_masked[~mask] = 0
    const float vn = mn ? dv : INFINITY;            // mat_masked[~mask] = inf
    if (vp > bp) { bp = vp; ip = j; }
    if (vn < bn) { bn = vn; in_ = j; }
    ap |= mp;
    an |= mn;
  }
  s_v[0][t] = bp; s_i[0][t] = ip; s_v[1][t] = bn; s_i[1][t] = in_;
  if (ap) atomicOr(&s_any[0], 1);
  if (an) atomicOr(&s_any[1], 1);
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) {
      if (s_v[0][t + o] > s_v[0][t] || (s_v[0][t + o] == s_v[0][t] && s_i[0][t + o] < s_i[0][t])) {
        s_v[0][t] = s_v[0][t + o]; s_i[0][t] = s_i[0][t + o];
      }
      if (s_v[1][t + o] < s_v[1][t] || (s_v[1][t + o] == s_v[1][t] && s_i[1][t + o] < s_i[1][t])) {
        s_v[1][t] = s_v[1][t + o]; s_i[1][t] = s_i[1][t + o];
      }
    }
    __syncthreads();
  }
  if (t == 0) {
    const bool keep = s_any[0] && s_any[1];
    trip[3 * i + 0] = keep ? i : -1;
    trip[3 * i + 1] = s_i[0][0] < n ? s_i[0][0] : 0;
    trip[3 * i + 2] = s_i[1][0] < n ? s_i[1][0] : 0;
    hp[i] = s_v[0][0];
    hn[i] = s_v[1][0];
  }
}

// out[0] loss, [1] num_triplets, [2] num_non_zero, [3] avg_embedding_norm, [4..6] mean/max/min hardest-positive
// distance, [7..9] mean/max/min hardest-negative distance (over ALL rows, as the reference does)
__global__ __launch_bounds__(256) void triplet_loss_kernel(const float* __restrict__ D, const int32_t* __restrict__ trip,
                                                           const float* __restrict__ hp, const float* __restrict__ hn,
                                                           const float* __restrict__ norms, int n, float margin,
                                                           float* __restrict__ li, float* __restrict__ out) {
  __shared__ float red[8][256];
  const int t = threadIdx.x;
  float sl = 0.f, nt = 0.f, nz = 0.f, sn = 0.f, sp = 0.f, mxp = -INFINITY, mnp = INFINITY, sng = 0.f, mxn = -INFINITY,
        mnn = INFINITY;
  for (int i = t; i < n; i += 256) {
    float l = 0.f;
    if (trip[3 * i] >= 0) {
      const int p = trip[3 * i + 1], q = trip[3 * i + 2];
      const float dap = D[(int64_t)i * n + p];
      const float dan = fminf(D[(int64_t)i * n + q], D[(int64_t)p * n + q]);     // swap=True
      l = fmaxf(dap - dan + margin, 0.f);
      nt += 1.f;
      if (l > 0.f) { nz += 1.f; sl += l; }
    }
    li[i] = l;
    sn += norms[i];
    sp += hp[i]; mxp = fmaxf(mxp, hp[i]); mnp = fminf(mnp, hp[i]);
    sng += hn[i]; mxn = fmaxf(mxn, hn[i]); mnn = fminf(mnn, hn[i]);
  }
  float v[10] = {sl, nt, nz, sn, sp, mxp, mnp, sng, mxn, mnn};
  // fixed-order tree reductions (deterministic); ops: sum sum sum sum sum max min sum max min
  for (int k = 0; k < 10; ++k) {
    __syncthreads();
    red[0][t] = v[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (t < o) {
        const float a = red[0][t], b = red[0][t + o];
        red[0][t] = (k == 5 || k == 8) ? fmaxf(a, b) : ((k == 6 || k == 9) ? fminf(a, b) : a + b);
      }
      __syncthreads();
    }
    v[k] = red[0][0];
  }
  if (t == 0) {
    out[0] = v[2] > 0.f ? v[0] / v[2] : 0.f;
    out[1] = v[1];
    out[2] = v[2];
    out[3] = v[3] / (float)n;
    out[4] = v[4] / (float)n; out[5] = v[5]; out[6] = v[6];
    out[7] = v[7] / (float)n; out[8] = v[8]; out[9] = v[9];
  }
}

// dLoss/dE[r][:]: row r gathers the contributions of every active triplet it takes part in (as anchor, positive or
// negative), in fixed triplet order => deterministic, no atomics.
__global__ __launch_bounds__(256) void triplet_grad_kernel(const float* __restrict__ e, const float* __restrict__ D,
                                                           const int32_t* __restrict__ trip,
                                                           const float* __restrict__ li,
                                                           const float* __restrict__ out, int n, int d,
                                                           float* __restrict__ grad) {
  const int r = blockIdx.x;
  const float cnt = out[2];
  const float w = cnt > 0.f ? 1.f / cnt : 0.f;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float g = 0.f;
    const float er = e[(int64_t)r * d + c];
    for (int i = 0; i < n; ++i) {
      if (trip[3 * i] < 0 || !(li[i] > 0.f)) continue;
      const int p = trip[3 * i + 1], q = trip[3 * i + 2];
      const float dap = D[(int64_t)i * n + p];
      const bool swapped = D[(int64_t)p * n + q] < D[(int64_t)i * n + q];
      const int x = swapped ? p : i;                 // the negative distance is d(x, q)
      const float dxq = D[(int64_t)x * n + q];
      // + d(i,p):  d/de_i = (e_i - e_p)/d, d/de_p = -(e_i - e_p)/d
      if (dap > 0.f) {
        if (r == i) g += w * (er - e[(int64_t)p * d + c]) / dap;
        if (r == p) g -= w * (e[(int64_t)i * d + c] - er) / dap;
      }
      // - d(x,q)
      if (dxq > 0.f) {
        if (r == x) g -= w * (er - e[(int64_t)q * d + c]) / dxq;
        if (r == q) g += w * (e[(int64_t)x * d + c] - er) / dxq;
      }
    }
    grad[(int64_t)r * d + c] = g;
  }
}

size_t triplet_loss_scratch_floats(int n) { return (size_t)n * n + 4 * (size_t)n + 64; }

int triplet_loss_forward(const float* emb, int n, int d, const uint8_t* pos, const uint8_t* neg, float margin,
                         float* out10, int32_t* triplets, float* grad, float* scratch, hipStream_t stream) {
  EGONN_REQUIRE(n >= 1 && d >= 1 && d <= 4096, EGONN_ERR_INVALID, "triplet loss: n=%d d=%d out of range", n, d);
  float* D = scratch;
  float* norms = D + (size_t)n * n;
  float* hp = norms + n;
  float* hn = hp + n;
  float* li = hn + n;
  hipLaunchKernelGGL(pdist_kernel, dim3(n), dim3(256), d * sizeof(float), stream, emb, n, d, D, norms);
  hipLaunchKernelGGL(mine_kernel, dim3(n), dim3(256), 0, stream, D, pos, neg, n, triplets, hp, hn);
  hipLaunchKernelGGL(triplet_loss_kernel, dim3(1), dim3(256), 0, stream, D, triplets, hp, hn, norms, n, margin, li,
                     out10);
  if (grad)
    hipLaunchKernelGGL(triplet_grad_kernel, dim3(n), dim3(256), 0, stream, emb, D, triplets, li, out10, n, d, grad);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------ local-head losses (models/loss_utils.py)
// KeypointLoss (:23-95) and CorrespondenceLoss (:108-139), driven per pair of scans by KeypointCorrLoss
// (models/loss.py:43-92), need three searches that the reference does on dense torch.cdist matrices:
//   * nearest keypoint of the other scan for every keypoint, both directions (probabilistic chamfer term, the classes of
//     the correspondence term),
//   * nearest cloud point for every keypoint (point-to-point term: torch.cdist(kp, pc) is keypoints x 50 k points),
//   * a row-wise softmax cross-entropy over the (kp1 x kp2) descriptor-similarity matrix.
// The kernels return indices and (for the CE) the loss rows + d loss / d logits; the differentiable tail that touches only
// (n,3)/(n,1) tensors stays with the host's autograd (egonn_amd/local_loss.py).

// nearest row of b (m,3) for every row of a (n,3), optionally transformed first: a' = R a + t (M: row-major 4x4, as
// misc/poses.py:68-76 apply_transform).  Distances in the difference form (exact for coincident points); ties: lowest index.
__global__ __launch_bounds__(256) void nn_search_kernel(const float* __restrict__ a, int32_t n, const float* __restrict__ M,
                                                         const float* __restrict__ b, int32_t m,
                                                         float* __restrict__ out_dist, int32_t* __restrict__ out_idx) {
  __shared__ float sb[1024 * 3];
  const int tid = threadIdx.x;
  const int32_t i = blockIdx.x * 256 + tid;
  float px = 0.f, py = 0.f, pz = 0.f;
  if (i < n) {
    const float x = a[3 * i], y = a[3 * i + 1], z = a[3 * i + 2];
    if (M) {   // pc @ m[:3,:3].T + m[:3,3], accumulated left to right like the matmul
      px = x * M[0] + y * M[1] + z * M[2] + M[3];
      py = x * M[4] + y * M[5] + z * M[6] + M[7];
      pz = x * M[8] + y * M[9] + z * M[10] + M[11];
    } else {
      px = x; py = y; pz = z;
    }
  }
  float best = INFINITY;
  int32_t bi = -1;
  for (int32_t j0 = 0; j0 < m; j0 += 1024) {
    const int32_t cnt = min(1024, m - j0);
    __syncthreads();
    for (int e = tid; e < cnt * 3; e += 256) sb[e] = b[(int64_t)j0 * 3 + e];
    __syncthreads();
    for (int32_t j = 0; j < cnt; ++j) {
      const float dx = px - sb[3 * j], dy = py - sb[3 * j + 1], dz = pz - sb[3 * j + 2];
      const float d2 = dx * dx + dy * dy + dz * dz;
      if (d2 < best) { best = d2; bi = j0 + j; }
    }
  }
  if (i < n) {
    out_dist[i] = sqrtf(best);
    out_idx[i] = bi;
  }
}
int nn_search(const float* a, int64_t n, const float* M, const float* b, int64_t m, float* out_dist, int32_t* out_idx,
              hipStream_t stream) {
  EGONN_REQUIRE(n >= 0 && m >= 1 && n < (1ll << 31) && m < (1ll << 31), EGONN_ERR_INVALID, "nn_search: bad sizes");
  if (n == 0) return EGONN_OK;
  hipLaunchKernelGGL(nn_search_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, a, (int32_t)n, M, b, (int32_t)m,
                     out_dist, out_idx);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// torch.min(d, dim=1) and torch.min(d, dim=0) of a dense (n,m) matrix (the reference hands KeypointLoss / CorrespondenceLoss a
// precomputed distance matrix): values + indices, ties: lowest index.  One wave per row; one thread per column.
__global__ __launch_bounds__(256) void matrix_min_rows_kernel(const float* __restrict__ d, int32_t n, int32_t m,
                                                               float* __restrict__ vmin, int32_t* __restrict__ imin) {
  const int lane = threadIdx.x & 63;
  const int32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  float best = INFINITY;
  int32_t bi = 0x7FFFFFFF;
  for (int32_t j = lane; j < m; j += 64) {
    const float v = d[(int64_t)r * m + j];
    if (v < best) { best = v; bi = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int32_t oi = __shfl_xor(bi, o, 64);
    if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) { vmin[r] = best; imin[r] = bi; }
}
__global__ void matrix_min_cols_kernel(const float* __restrict__ d, int32_t n, int32_t m, float* __restrict__ vmin,
                                       int32_t* __restrict__ imin) {
  const int32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= m) return;
  float best = INFINITY;
  int32_t bi = 0;
  for (int32_t r = 0; r < n; ++r) {
    const float v = d[(int64_t)r * m + c];
    if (v < best) { best = v; bi = r; }
  }
  vmin[c] = best;
  imin[c] = bi;
}
int matrix_min(const float* d, int64_t n, int64_t m, float* row_min, int32_t* row_idx, float* col_min, int32_t* col_idx,
               hipStream_t stream) {
  EGONN_REQUIRE(n >= 1 && m >= 1 && n < (1ll << 31) && m < (1ll << 31), EGONN_ERR_INVALID, "matrix_min: bad sizes");
  hipLaunchKernelGGL(matrix_min_rows_kernel, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, stream, d, (int32_t)n, (int32_t)m,
                     row_min, row_idx);
  hipLaunchKernelGGL(matrix_min_cols_kernel, dim3((unsigned)cdiv(m, 256)), dim3(256), 0, stream, d, (int32_t)n, (int32_t)m,
                     col_min, col_idx);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// Row-wise softmax cross-entropy (torch.nn.CrossEntropyLoss on the similarity matrix, loss_utils.py:127-128): for every
// row r with target[r] >= 0:  loss[r] = logsumexp(logits[r]) - logits[r][target[r]];  dlogits[r] = softmax(logits[r]) -
// onehot(target[r]).  Rows with target < 0 (keypoints without a correspondence, :124-125) give 0.  One wave per row,
// max-shifted exponentials, fixed-order reductions.  argmax[r] (ties: lowest index) serves the 'matching_descriptors' metric.
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ logits, int32_t n, int32_t m,
                                                          const int32_t* __restrict__ target, float* __restrict__ loss,
                                                          int32_t* __restrict__ argmax, float* __restrict__ dlogits) {
  const int lane = threadIdx.x & 63;
  const int32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const float* row = logits + (int64_t)r * m;
  const int32_t t = target[r];
  float mx = -INFINITY;
  int32_t mi = 0x7FFFFFFF;
  for (int32_t j = lane; j < m; j += 64) {
    const float v = row[j];
    if (v > mx) { mx = v; mi = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(mx, o, 64);
    const int32_t oi = __shfl_xor(mi, o, 64);
    if (ov > mx || (ov == mx && oi < mi)) { mx = ov; mi = oi; }
  }
  float se = 0.f;
  for (int32_t j = lane; j < m; j += 64) se += expf(row[j] - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, 64);
  if (lane == 0) {
    loss[r] = (t >= 0) ? (logf(se) + mx - row[t]) : 0.f;
    argmax[r] = mi;
  }
  if (dlogits) {
    const float inv = 1.f / se;
    for (int32_t j = lane; j < m; j += 64)
      dlogits[(int64_t)r * m + j] = (t >= 0) ? (expf(row[j] - mx) * inv - (j == t ? 1.f : 0.f)) : 0.f;
  }
}
int softmax_ce(const float* logits, int64_t n, int64_t m, const int32_t* target, float* loss, int32_t* argmax, float* dlogits,
               hipStream_t stream) {
  EGONN_REQUIRE(n >= 0 && m >= 1 && n < (1ll << 31) && m < (1ll << 31), EGONN_ERR_INVALID, "softmax_ce: bad sizes");
  if (n == 0) return EGONN_OK;
  hipLaunchKernelGGL(softmax_ce_kernel, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, stream, logits, (int32_t)n, (int32_t)m,
                     target, loss, argmax, dlogits);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

}  // namespace egonn
