// Scan ingest on the device — the step right before the descriptor path: the raw `.bin` payload
// (float32 x,y,z,reflectance per return: datasets/mulran/mulran_raw.py:19-25, datasets/kitti/kitti_raw.py:16-22) of a
// whole batch is filtered exactly like PointCloudLoader.__call__ (misc/point_clouds.py:95-111)
//     mask = np.all(np.isclose(pc, 0), axis=1); pc = pc[~mask]          (|v| <= 1e-8 on all three coordinates)
//     pc = pc[pc[:, 2] > ground_plane_level]                             (NaN z fails the test, as in numpy)
// and compacted in the original point order (stable) into the (n, 3) array egonn_voxelize consumes, together with
// the per-scan offsets of the survivors.  Three small launches: per-block counts, one-block scan, scatter.
#include "common.h"
#include "kernels.h"

namespace egonn {

static constexpr int ING_BLOCK = 1024;   // points per workgroup (256 threads x 4 rounds)

__device__ static inline bool keep_point(const float* __restrict__ raw, int64_t i, int stride, int remove_zero,
                                         int remove_ground, float ground, float& x, float& y, float& z) {
  x = raw[i * stride + 0];
  y = raw[i * stride + 1];
  z = raw[i * stride + 2];
  bool keep = true;
  if (remove_zero && fabsf(x) <= 1e-8f && fabsf(y) <= 1e-8f && fabsf(z) <= 1e-8f) keep = false;
  if (remove_ground && !(z > ground)) keep = false;
  return keep;
}

__global__ __launch_bounds__(256) void ingest_count_kernel(const float* __restrict__ raw, int64_t n, int stride,
                                                          int remove_zero, int remove_ground, float ground,
                                                          const int64_t* __restrict__ raw_off, int batch,
                                                          int32_t* __restrict__ block_cnt) {
  __shared__ int s_w[4];
  const int t = threadIdx.x;
  n = min(n, raw_off[batch]);               // n is a capacity: the rows in use are known on the device only
  int c = 0;
  for (int r = 0; r < 4; ++r) {
    const int64_t i = (int64_t)blockIdx.x * ING_BLOCK + r * 256 + t;
    float x, y, z;
    const bool k = i < n && keep_point(raw, i, stride, remove_zero, remove_ground, ground, x, y, z);
    c += __popcll(__ballot(k));
  }
  if ((t & 63) == 0) s_w[t >> 6] = c;       // every lane of a wave holds the wave's count
  __syncthreads();
  if (t == 0) block_cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// exclusive scan of block_cnt (in place) by ONE workgroup; total -> block_cnt[nblk]
__global__ __launch_bounds__(1024) void ingest_scan_kernel(int32_t* __restrict__ block_cnt, int32_t nblk) {
  __shared__ int32_t s_w[16];
  __shared__ int32_t s_carry;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t == 0) s_carry = 0;
  __syncthreads();
  for (int32_t base = 0; base < nblk; base += 1024) {
    const int32_t i = base + t;
    const int32_t v = i < nblk ? block_cnt[i] : 0;
    int32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t u = __shfl_up(inc, o, 64);
      if (lane >= o) inc += u;
    }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int32_t woff = 0;
    for (int j = 0; j < w; ++j) woff += s_w[j];
    const int32_t carry = s_carry;
    if (i < nblk) block_cnt[i] = carry + woff + inc - v;
    __syncthreads();
    if (t == 1023) s_carry = carry + woff + inc;
    __syncthreads();
  }
  if (t == 0) block_cnt[nblk] = s_carry;
}

__global__ __launch_bounds__(256) void ingest_scatter_kernel(const float* __restrict__ raw, int64_t n, int stride,
                                                            int remove_zero, int remove_ground, float ground,
                                                            const int32_t* __restrict__ block_pre,
                                                            const int64_t* __restrict__ raw_off, int batch,
                                                            float* __restrict__ out) {
  __shared__ int s_w[4];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  n = min(n, raw_off[batch]);
  int32_t pos = block_pre[blockIdx.x];
  for (int r = 0; r < 4; ++r) {
    const int64_t i = (int64_t)blockIdx.x * ING_BLOCK + r * 256 + t;
    float x = 0.f, y = 0.f, z = 0.f;
    const bool k = i < n && keep_point(raw, i, stride, remove_zero, remove_ground, ground, x, y, z);
    const uint64_t bal = __ballot(k);
    if (lane == 0) s_w[w] = __popcll(bal);
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < w) woff += s_w[j];
      tot += s_w[j];
    }
    if (k) {
      const int64_t o = (int64_t)pos + woff + __popcll(bal & ((1ull << lane) - 1ull));
      out[o * 3 + 0] = x;
      out[o * 3 + 1] = y;
      out[o * 3 + 2] = z;
    }
    pos += tot;
    __syncthreads();
  }
}

// survivors with raw index < raw_off[b]  (raw_off on the device, B+1 entries); one wave per entry
__global__ __launch_bounds__(64) void ingest_offsets_kernel(const float* __restrict__ raw, int64_t n, int stride,
                                                           int remove_zero, int remove_ground, float ground,
                                                           const int32_t* __restrict__ block_pre,
                                                           const int64_t* __restrict__ raw_off, int nb1,
                                                           int64_t* __restrict__ new_off) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= nb1) return;
  const int64_t e = raw_off[b];
  const int64_t blk = e / ING_BLOCK;
  int c = 0;
  for (int64_t i0 = blk * ING_BLOCK; i0 < e; i0 += 64) {
    const int64_t i = i0 + lane;
    float x, y, z;
    const bool k = i < e && keep_point(raw, i, stride, remove_zero, remove_ground, ground, x, y, z);
    c += __popcll(__ballot(k));
  }
  if (lane == 0) new_off[b] = (int64_t)block_pre[blk] + c;
}

size_t ingest_scratch_ints(int64_t n) { return (size_t)cdiv(n, ING_BLOCK) + 2; }

int ingest_filter(const float* raw, int64_t n, int stride, const int64_t* raw_off_dev, int batch, int remove_zero,
                  int remove_ground, float ground, float* out_xyz, int64_t* new_off_dev, int32_t* scratch,
                  size_t scratch_ints, hipStream_t stream) {
  EGONN_REQUIRE(raw && out_xyz && raw_off_dev && new_off_dev && scratch && n >= 0 && n < (1ll << 31) && batch >= 1 &&
                    (stride == 3 || stride == 4),
                EGONN_ERR_INVALID, "ingest: bad arguments (n=%lld stride=%d)", (long long)n, stride);
  const int64_t nblk = cdiv(n, ING_BLOCK);
  EGONN_REQUIRE(scratch_ints >= (size_t)nblk + 2, EGONN_ERR_INVALID, "ingest: scratch too small");
  if (nblk > 0) {
    hipLaunchKernelGGL(ingest_count_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, raw, n, stride, remove_zero,
                       remove_ground, ground, raw_off_dev, batch, scratch);
  }
  hipLaunchKernelGGL(ingest_scan_kernel, dim3(1), dim3(1024), 0, stream, scratch, (int32_t)nblk);
  if (nblk > 0)
    hipLaunchKernelGGL(ingest_scatter_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, raw, n, stride, remove_zero,
                       remove_ground, ground, scratch, raw_off_dev, batch, out_xyz);
  hipLaunchKernelGGL(ingest_offsets_kernel, dim3((unsigned)(batch + 1)), dim3(64), 0, stream, raw, n, stride,
                     remove_zero, remove_ground, ground, scratch, raw_off_dev, batch + 1, new_off_dev);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

}  // namespace egonn
