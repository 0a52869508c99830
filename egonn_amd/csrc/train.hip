// Training-mode kernels (BASELINE configs[3]: the sharded training step).
//
// The backward pass of the sparse convolutions reuses the forward machinery wherever the map is its own
// transpose:  dgrad of a k=3 convolution is the same convolution with W'[k] = W[26-k]^T on the same table
// (nbr[o][k] = j  <=>  nbr[j][26-k] = o), dgrad of the k=2,s=2 convolution is the transposed-convolution kernel
// and vice versa — those are composed on the host side (egonn_amd/train.py) from egonn_conv /
// egonn_conv_transpose.  What is new here:
//   * weight gradients  dW[k] = sum over pairs (o, j = nbr[o][k]) of in[j]^T (x) dout[o]   (pair-compacted, two-stage,
//     deterministic) for table-driven convolutions, dense layers (identity map) and the k=5 input layer;
//   * batch-statistics MinkowskiBatchNorm (= nn.BatchNorm1d over all rows, reference models/minkgl.py:102,107 in
//     train mode) forward / backward reductions and the element-wise passes around them;
//   * backward of the ECA gate/residual tail (layers/eca_block.py:66-73), of the per-sample average pooling and of
//     GeM (layers/pooling.py:82-86).
// Per-channel / per-sample vectors (C or B x C values) are combined on the host side with tiny tensor ops; for
// SyncBN the (sum, count) vectors are all-reduced over RCCL between the reduction and the element-wise kernel.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace egonn {

// ------------------------------------------------------------------------------------------- weight gradient
// grid (chunks, K, output tiles); block = (TCI/4)*(TCO/4) threads, each owns a 4x4 register tile of dW[k].
// Rows of the chunk are scanned NT at a time; valid pairs are compacted with ballot/popcount and then consumed PB
// at a time through LDS (in rows [PB][TCI], dout rows [PB][TCO]).
template <int TCI, int TCO>
__global__ __launch_bounds__((TCI / 4) * (TCO / 4)) void wgrad_kernel(
    const float* __restrict__ in, const float* __restrict__ dout, const int32_t* __restrict__ nbr, int32_t n_out, int K,
    int cin, int cout, int32_t rows_per_chunk, float* __restrict__ partial) {
  constexpr int NT = (TCI / 4) * (TCO / 4);
  constexpr int NW = (NT + 63) / 64;
  constexpr int PB = 32;
  __shared__ int2 s_pair[NT];
  __shared__ __attribute__((aligned(16))) float s_a[PB][TCI];
  __shared__ __attribute__((aligned(16))) float s_b[PB][TCO];
  __shared__ int s_wcnt[NW];
  const int t = threadIdx.x, k = blockIdx.y, chunk = blockIdx.x;
  const int tiles_co = (cout + TCO - 1) / TCO;
  const int ci0 = ((int)blockIdx.z / tiles_co) * TCI, co0 = ((int)blockIdx.z % tiles_co) * TCO;
  const int ty = t / (TCO / 4), tx = t % (TCO / 4);
  const int lane = t & 63, w = t >> 6;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const int32_t r0 = chunk * rows_per_chunk;
  const int32_t r1 = (int32_t)min((int64_t)n_out, (int64_t)r0 + rows_per_chunk);
  for (int32_t base = r0; base < r1; base += NT) {
    const int32_t o = base + t;
    int32_t j = -1;
    if (o < r1) j = nbr ? nbr[(int64_t)o * K + k] : o;
    const bool v = j >= 0;
    const uint64_t bal = __ballot(v);
    if (lane == 0) s_wcnt[w] = __popcll(bal);
    __syncthreads();
    int woff = 0, cnt = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      if (i < w) woff += s_wcnt[i];
      cnt += s_wcnt[i];
    }
    if (v) s_pair[woff + __popcll(bal & ((1ull << lane) - 1ull))] = make_int2(j, o);
    __syncthreads();
    for (int p0 = 0; p0 < cnt; p0 += PB) {
      for (int e = t; e < PB * TCI; e += NT) {
        const int pr = e / TCI, c = e % TCI;
        float val = 0.f;
        if (p0 + pr < cnt && ci0 + c < cin) val = in[(int64_t)s_pair[p0 + pr].x * cin + ci0 + c];
        s_a[pr][c] = val;
      }
      for (int e = t; e < PB * TCO; e += NT) {
        const int pr = e / TCO, c = e % TCO;
        float val = 0.f;
        if (p0 + pr < cnt && co0 + c < cout) val = dout[(int64_t)s_pair[p0 + pr].y * cout + co0 + c];
        s_b[pr][c] = val;
      }
      __syncthreads();
#pragma unroll 8
      for (int pr = 0; pr < PB; ++pr) {
        const float4 a = *reinterpret_cast<const float4*>(&s_a[pr][ty * 4]);
        const float4 b = *reinterpret_cast<const float4*>(&s_b[pr][tx * 4]);
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) acc[i][jj] = fmaf(av[i], bv[jj], acc[i][jj]);
      }
      __syncthreads();
    }
  }
  float* dst = partial + ((int64_t)chunk * K + k) * cin * cout;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ci = ci0 + ty * 4 + i;
    if (ci >= cin) continue;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int co = co0 + tx * 4 + jj;
      if (co < cout) dst[(int64_t)ci * cout + co] = acc[i][jj];
    }
  }
}

// MFMA version for the sparse-conv channel plans: dW[k] (CIN x COUT) = A^T B with A = gathered input rows (pairs x CIN),
// B = grad_out rows (pairs x COUT); v_mfma_f32_16x16x4_f32 with M = ci, N = co, K = 4 pairs per instruction.  Four
// waves tile the CIN x COUT output (WM x WN waves, each (TM/WM) x (TN/WN) tiles of 16 x 16, accumulators in registers);
// pairs are compacted like in wgrad_kernel and staged 32 at a time in LDS (row stride +16 floats: the four
// lane groups of a fragment read four consecutive rows -> bank offsets 0/16, conflict free).
typedef float wg_f32x4 __attribute__((ext_vector_type(4)));
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(const float* __restrict__ in, const float* __restrict__ dout,
                                                        const int32_t* __restrict__ nbr, int32_t n_out, int K,
                                                        int32_t rows_per_chunk, float* __restrict__ partial,
                                                        const int32_t* __restrict__ rg_perm, const int32_t* __restrict__ rg_snbr,
                                                        const int32_t* __restrict__ rg_meta) {
  constexpr int TM = CIN / 16, TN = COUT / 16;
  constexpr int WN = 2, WM = 2;
  constexpr int TMW = TM / WM, TNW = TN / WN;
  constexpr int PB = 32, LDA = CIN + 16, LDB = COUT + 16;
  static_assert(TM % WM == 0 && TN % WN == 0, "bad wgrad tiling");
  __shared__ int2 s_pair[256];
  __shared__ __attribute__((aligned(16))) float s_a[PB * LDA];
  __shared__ __attribute__((aligned(16))) float s_b[PB * LDB];
  __shared__ int s_wcnt[4];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, k = blockIdx.y, chunk = blockIdx.x;
  const int wm = w / WN, wn = w % WN;
  const int l15 = lane & 15, g4 = lane >> 4;
  wg_f32x4 acc[TMW][TNW];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < TNW; ++j) acc[i][j] = (wg_f32x4){0.f, 0.f, 0.f, 0.f};
  // rg_perm != null: the pairs come from the row-group form of the map (rowgroup.hip) — slot e = 16 g + s holds output row
  // perm[e] and input row snbr[(g K + k) 16 + s]: one 64-byte line per (group, offset) where the plain table costs every
  // (chunk, offset) workgroup a 108-byte-strided column (each line of the table fetched 27 times); n_out counts slots then
  if (rg_perm) n_out = min(n_out, rg_meta[0] * 16);
  const int32_t r0 = chunk * rows_per_chunk;
  const int32_t r1 = (int32_t)min((int64_t)n_out, (int64_t)r0 + rows_per_chunk);
  for (int32_t base = r0; base < r1; base += 256) {
    int32_t o = base + t;
    int32_t j = -1;
    if (o < r1) {
      if (rg_perm) {
        const int32_t e = o;
        o = rg_perm[e];
        if (o >= 0) j = rg_snbr[((int64_t)(e >> 4) * K + k) * 16 + (e & 15)];
      } else {
        j = nbr ? nbr[(int64_t)o * K + k] : o;
      }
    }
    const bool v = j >= 0;
    const uint64_t bal = __ballot(v);
    if (lane == 0) s_wcnt[w] = __popcll(bal);
    __syncthreads();
    int woff = 0, cnt = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < w) woff += s_wcnt[i];
      cnt += s_wcnt[i];
    }
    if (v) s_pair[woff + __popcll(bal & ((1ull << lane) - 1ull))] = make_int2(j, o);
    __syncthreads();
    // the rows of batch p0 + PB are requested (into registers) before the MFMAs of batch p0: the gather latency (a global round
    // trip, ~2 us on a busy chip) used to sit between every two batches of at most 0.1-1 us of matrix work
    constexpr int NA = (PB * (CIN / 4) + 255) / 256, NB = (PB * (COUT / 4) + 255) / 256;
    float4 ra[NA], rb[NB];
    auto fetch = [&](int p0) {
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        const int e = t + 256 * q, pr = e / (CIN / 4), c4 = e % (CIN / 4);
        ra[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < PB * (CIN / 4) && p0 + pr < cnt) ra[q] = *reinterpret_cast<const float4*>(in + (int64_t)s_pair[p0 + pr].x * CIN + c4 * 4);
      }
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        const int e = t + 256 * q, pr = e / (COUT / 4), c4 = e % (COUT / 4);
        rb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < PB * (COUT / 4) && p0 + pr < cnt) rb[q] = *reinterpret_cast<const float4*>(dout + (int64_t)s_pair[p0 + pr].y * COUT + c4 * 4);
      }
    };
    if (cnt > 0) fetch(0);
    for (int p0 = 0; p0 < cnt; p0 += PB) {
#pragma unroll
      for (int q = 0; q < NA; ++q) {
        const int e = t + 256 * q, pr = e / (CIN / 4), c4 = e % (CIN / 4);
        if (e < PB * (CIN / 4)) *reinterpret_cast<float4*>(&s_a[pr * LDA + c4 * 4]) = ra[q];
      }
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        const int e = t + 256 * q, pr = e / (COUT / 4), c4 = e % (COUT / 4);
        if (e < PB * (COUT / 4)) *reinterpret_cast<float4*>(&s_b[pr * LDB + c4 * 4]) = rb[q];
      }
      __syncthreads();
      if (p0 + PB < cnt) fetch(p0 + PB);
#pragma unroll
      for (int sst = 0; sst < PB / 4; ++sst) {
        float a[TMW], b[TNW];
#pragma unroll
        for (int i = 0; i < TMW; ++i) a[i] = s_a[(4 * sst + g4) * LDA + (wm * TMW + i) * 16 + l15];
#pragma unroll
        for (int jn = 0; jn < TNW; ++jn) b[jn] = s_b[(4 * sst + g4) * LDB + (wn * TNW + jn) * 16 + l15];
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
          for (int jn = 0; jn < TNW; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[jn], acc[i][jn], 0, 0, 0);
      }
      __syncthreads();
    }
  }
  float* dst = partial + ((int64_t)chunk * K + k) * CIN * COUT;
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int jn = 0; jn < TNW; ++jn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = (wm * TMW + i) * 16 + 4 * g4 + r, co = (wn * TNW + jn) * 16 + l15;
        dst[(int64_t)ci * COUT + co] = acc[i][jn][r];
      }
}

__global__ void sum_partials_kernel(const float* __restrict__ partial, int chunks, int64_t size, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= size) return;
  double s = 0.0;
  for (int ch = 0; ch < chunks; ++ch) s += (double)partial[(int64_t)ch * size + i];
  out[i] = (float)s;
}

// The same sum with the chunks spread over 16 lanes per output (many chunks, few outputs: the conv0 / k=2 / 32-channel weight
// gradients — one thread per output walked 256-512 chunks one after the other: 60-124 us per call): block = 16 outputs x 16
// chunk lanes, every lane sums its chunks in ascending order (fp64), the 16 lane sums are added in lane order.
__global__ __launch_bounds__(256) void sum_partials16_kernel(const float* __restrict__ partial, int chunks, int64_t size,
                                                            float* __restrict__ out) {
  __shared__ double red[16][17];
  const int o = threadIdx.x & 15, cl = threadIdx.x >> 4;
  const int64_t i = (int64_t)blockIdx.x * 16 + o;
  double s = 0.0;
  if (i < size) {
    int ch = cl;
    for (; ch + 48 < chunks; ch += 64) {                    // four loads in flight
      const float v0 = partial[(int64_t)ch * size + i], v1 = partial[(int64_t)(ch + 16) * size + i];
      const float v2 = partial[(int64_t)(ch + 32) * size + i], v3 = partial[(int64_t)(ch + 48) * size + i];
      s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
    }
    for (; ch < chunks; ch += 16) s += (double)partial[(int64_t)ch * size + i];
  }
  red[cl][o] = s;
  __syncthreads();
  if (cl == 0 && i < size) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][o];
    out[i] = (float)t;
  }
}
static void launch_sum_partials(const float* partial, int chunks, int64_t size, float* out, hipStream_t stream) {
  if (chunks >= 64 && size <= (1 << 17))
    hipLaunchKernelGGL(sum_partials16_kernel, dim3((unsigned)cdiv(size, 16)), dim3(256), 0, stream, partial, chunks, size, out);
  else
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)cdiv(size, 256)), dim3(256), 0, stream, partial, chunks, size, out);
}

int conv_wgrad(const float* in, const float* dout, const int32_t* nbr, int64_t n_out, int K, int cin, int cout,
               float* dW, float* scratch, size_t scratch_floats, hipStream_t stream, const RowGroups* rg) {
  const int64_t size = (int64_t)K * cin * cout;
  if (n_out == 0) {
    HIP_CHECK(hipMemsetAsync(dW, 0, (size_t)size * 4, stream));
    return EGONN_OK;
  }
  EGONN_REQUIRE(scratch && scratch_floats >= (size_t)size, EGONN_ERR_INVALID,
                "wgrad: scratch of %zu floats is smaller than one kernel (%lld)", scratch_floats, (long long)size);
  const bool mfma = (cin == 32 && (cout == 32 || cout == 64)) || (cin == 64 && (cout == 64 || cout == 128)) ||
                    (cin == 128 && cout == 128);
  if (mfma) {
    const bool use_rg = rg && rg->built && rg->K == K && K > 1;
    if (use_rg) n_out = (int64_t)rg->cap_groups * 16;      // slots (the kernel clips to the groups in use)
    // (chunk, offset) workgroups: enough of them to fill the chip (a chunk is a chain of dependent 256-row slabs: the small
    // levels ran 3-17 workgroups for 72-79 us per call), never fewer than 256 rows per chunk
    int64_t chunks = std::max<int64_t>(1, 2048 / (int64_t)K);
    chunks = std::min<int64_t>(chunks, cdiv(n_out, 256));
    chunks = std::max<int64_t>(1, std::min<int64_t>(chunks, (int64_t)(scratch_floats / (size_t)size)));
    const int32_t rpc = (int32_t)cdiv(n_out, chunks);
    chunks = cdiv(n_out, rpc);
    const dim3 grid((unsigned)chunks, (unsigned)K);
#define EGONN_WGRAD_CASE(CI, CO)                                                                                 \
  if (cin == CI && cout == CO)                                                                                   \
    hipLaunchKernelGGL((wgrad_mfma_kernel<CI, CO>), grid, dim3(256), 0, stream, in, dout, nbr, (int32_t)n_out, K, rpc, \
                       scratch, use_rg ? rg->perm : nullptr, use_rg ? rg->snbr : nullptr, use_rg ? rg->meta : nullptr);
    EGONN_WGRAD_CASE(32, 32)
    EGONN_WGRAD_CASE(32, 64)
    EGONN_WGRAD_CASE(64, 64)
    EGONN_WGRAD_CASE(64, 128)
    EGONN_WGRAD_CASE(128, 128)
#undef EGONN_WGRAD_CASE
    launch_sum_partials(scratch, (int)chunks, size, dW, stream);
    HIP_CHECK(hipGetLastError());
    return EGONN_OK;
  }
  const bool small = cin <= 32 && cout <= 32;
  const int T = small ? 32 : 64, NT = small ? 64 : 256;
  const int tiles = (int)(cdiv(cin, T) * cdiv(cout, T));
  int64_t chunks = std::max<int64_t>(1, 4096 / ((int64_t)K * tiles));
  chunks = std::min<int64_t>(chunks, cdiv(n_out, NT));    // (one slab of NT rows per chunk at least: a chunk is a serial chain of slabs)
  chunks = std::min<int64_t>(chunks, (int64_t)(scratch_floats / (size_t)size));
  chunks = std::max<int64_t>(chunks, 1);
  const int32_t rpc = (int32_t)cdiv(n_out, chunks);
  chunks = cdiv(n_out, rpc);
  const dim3 grid((unsigned)chunks, (unsigned)K, (unsigned)tiles);
  if (small)
    hipLaunchKernelGGL((wgrad_kernel<32, 32>), grid, dim3(64), 0, stream, in, dout, nbr, (int32_t)n_out, K, cin, cout, rpc,
                       scratch);
  else
    hipLaunchKernelGGL((wgrad_kernel<64, 64>), grid, dim3(256), 0, stream, in, dout, nbr, (int32_t)n_out, K, cin, cout,
                       rpc, scratch);
  launch_sum_partials(scratch, (int)chunks, size, dW, stream);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------ weight gradient of the k=5, Cin=1 input layer
// dW[k][c] = sum over voxels o with an occupied neighbour j at offset k of f[j] * dout[o][c].  No kernel map: the
// neighbour test is the forward kernel's (lookup table -> block slot/bit -> occupancy mask of the plan's per-block
// table).  Block = 8 row lanes x 32 channels; thread (c, g) owns offsets k = g, g+8, ...
static constexpr int C0_ROWS = 8;
__global__ __launch_bounds__(256) void conv0_wgrad_kernel(const float* __restrict__ feat, const float* __restrict__ dout,
                                                         const uint64_t* __restrict__ keys, const int32_t* __restrict__ g0,
                                                         const uint64_t* __restrict__ t2m, const int32_t* __restrict__ t2s,
                                                         const uint16_t* __restrict__ lut, int32_t n0,
                                                         int32_t rows_per_chunk, float* __restrict__ partial) {
  __shared__ uint16_t s_lut[64 * 128];
  __shared__ uint64_t s_m[C0_ROWS][27];
  __shared__ int32_t s_s[C0_ROWS][27];
  __shared__ int s_lk[C0_ROWS];
  __shared__ float s_d[C0_ROWS][32];
  const int t = threadIdx.x, c = t & 31, g = t >> 5;
  for (int e = t; e < 64 * 128 / 2; e += 256)
    reinterpret_cast<uint32_t*>(s_lut)[e] = reinterpret_cast<const uint32_t*>(lut)[e];
  float acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  const int32_t r0 = blockIdx.x * rows_per_chunk, r1 = min(n0, r0 + rows_per_chunk);
  for (int32_t base = r0; base < r1; base += C0_ROWS) {
    __syncthreads();
    if (t < C0_ROWS * 27) {
      const int r = t / 27, s = t % 27;
      const int32_t o = base + r;
      uint64_t m = 0;
      int32_t st = 0;
      if (o < r1) {
        const int32_t blk = g0[o];
        m = t2m[(int64_t)blk * 27 + s];
        st = t2s[(int64_t)blk * 27 + s];
      }
      s_m[r][s] = m;
      s_s[r][s] = st;
    }
    if (t < C0_ROWS) s_lk[t] = (base + t < r1) ? (int)(keys[base + t] & 63) : 0;
    {
      const int r = t >> 5;
      s_d[r][c] = (base + r < r1) ? dout[(int64_t)(base + r) * 32 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < C0_ROWS; ++r) {
      const int lk = s_lk[r];
      const float d = s_d[r][c];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int k = g + 8 * q;
        const uint32_t e = s_lut[lk * 128 + k];          // (8 * block slot) << 6 | bit; slot 27 = padding offset
        const int slot = e >> 9, bit = e & 63;
        if (slot >= 27) continue;
        const uint64_t m = s_m[r][slot];
        if ((m >> bit) & 1ull) {
          float f = 1.f;
          if (feat) f = feat[s_s[r][slot] + __popcll(m & ((1ull << bit) - 1ull))];
          acc[q] = fmaf(f, d, acc[q]);
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int k = g + 8 * q;
    if (k < 125) partial[((int64_t)blockIdx.x * 125 + k) * 32 + c] = acc[q];
  }
}


// Unit input features (feat == NULL: what the reference always feeds): dW[k][c] = sum over voxels v with an occupied
// neighbour at offset k of dout[v][c] is the matrix product occ^T (128 x N, entries 0 / 1) @ dout (N x 32) and runs on
// v_mfma_f32_16x16x32_bf16: the occupancy operand is exact in bf16, dout is split exactly into three bf16 parts (as the forward
// kernel conv0_k5_unit_kernel splits W), fp32 accumulation.  Per tile of 32 voxels a wave
//   * builds every voxel's 128-bit occupancy vector the way the forward kernel builds its operand (lookup table -> block slot /
//     bit -> mask test; lane = (voxel, 8 consecutive offsets per 16-byte table read)) and parks it in LDS (16 B per voxel);
//   * reads the vectors of its 8 contraction voxels back (broadcast reads) and expands bit (offset) into bf16 0 / 1 per M tile;
//   * stages the dout tile (32 x 32 fp32, coalesced) through LDS, reads it transposed and splits it hi / mid / lo;
//   * issues 8 (offset tiles) x 2 (channel tiles) x 3 (parts) MFMAs into 64 accumulator registers that live for the whole kernel.
// 1.58 ms -> ~0.05 ms at the 830 k voxels of a 32-scan training step (the plain kernel did 3.3 G predicated FMAs).
// Deterministic: tiles are dealt to the waves in a fixed order, the partials are summed in fixed order (sum_partials_kernel).
typedef float f32x4_t4 __attribute__((ext_vector_type(4)));
typedef float f32x2_t4 __attribute__((ext_vector_type(2)));
typedef short bf16x8_t4 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t4 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void conv0_wgrad_unit_kernel(const float* __restrict__ dout, const uint64_t* __restrict__ vkeys,
                                                              const int32_t* __restrict__ g0, const uint64_t* __restrict__ t2m,
                                                              const uint16_t* __restrict__ lut, int32_t nvox, int32_t n2,
                                                              float* __restrict__ partial) {
  constexpr int MS = 29;                                   // 27 neighbour blocks + the all-zero mask (slot 27)
  constexpr int LUT_STRIDE = 136;
  __shared__ uint64_t s_m[4][16][MS];
  __shared__ __attribute__((aligned(16))) uint16_t s_lut[64 * LUT_STRIDE];
  __shared__ __attribute__((aligned(16))) uint8_t s_occ[4][32][16];       // per wave: 32 voxels x 128 occupancy bits
  __shared__ __attribute__((aligned(16))) float s_d[4][32][36];           // per wave: dout tile, rows padded to 36 floats
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g4 = lane >> 4;
  for (int e = tid; e < 64 * 128 / 2; e += 256) {
    const int r = e >> 6, c = e & 63;
    reinterpret_cast<uint32_t*>(s_lut)[r * (LUT_STRIDE / 2) + c] = reinterpret_cast<const uint32_t*>(lut)[e];
  }
  if (tid < 64) s_m[tid >> 4][tid & 15][27] = 0ull;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t m_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(t2m), 0, (int)((uint32_t)n2 * 27u * 8u), 0x00020000);
  const __amdgpu_buffer_rsrc_t g_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(g0), 0, (int)((uint32_t)nvox * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t k_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(vkeys), 0, (int)((uint32_t)nvox * 8u), 0x00020000);
  const __amdgpu_buffer_rsrc_t d_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dout), 0, (int)((uint32_t)nvox * 128u), 0x00020000);
  f32x4_t4 acc[8][2];
#pragma unroll
  for (int t = 0; t < 8; ++t) acc[t][0] = acc[t][1] = (f32x4_t4){0.f, 0.f, 0.f, 0.f};
  const int32_t ntiles = (nvox + 31) >> 5;
  constexpr int NE = (16 * 27 + 63) / 64;
  for (int32_t tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
    // ---- dout tile -> LDS (4 coalesced 1 KB loads; rows beyond nvox read zeros)
    f32x4_t4 dv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      dv[q] = __builtin_bit_cast(f32x4_t4, __builtin_amdgcn_raw_buffer_load_b128(d_rsrc, (tile * 32 + q * 8 + (lane >> 3)) * 128 + (lane & 7) * 16, 0, 0));
    // ---- occupancy vectors of the 32 voxels, 16 at a time (the forward kernel's operand construction)
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      const uint32_t r = (uint32_t)tile * 32u + (uint32_t)h * 16u + (uint32_t)l15;
      const int32_t gblk = __builtin_amdgcn_raw_buffer_load_b32(g_rsrc, (int)(r * 4u), 0, 0);
      const uint32_t lk = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(k_rsrc, (int)(r * 8u), 0, 0) & 63u;
      uint64_t pm[NE];
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int idx = lane + 64 * e;
        const int row = idx / 27, slot = idx - row * 27;
        const int32_t gg = __shfl(gblk, row & 15, 64);
        const bool ok = idx < 16 * 27 && (uint32_t)tile * 32u + (uint32_t)h * 16u + (uint32_t)row < (uint32_t)nvox;
        const uint32_t ent = ok ? (uint32_t)gg * 27u + (uint32_t)slot : 0x3FFFFFFFu;          // beyond the data: zeros
        const auto m2 = __builtin_amdgcn_raw_buffer_load_b64(m_rsrc, (int)(ent * 8u), 0, 0);
        pm[e] = ((uint64_t)m2[1] << 32) | (uint64_t)m2[0];
      }
      __builtin_amdgcn_wave_barrier();                     // (the previous half's mask reads are done)
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int idx = lane + 64 * e;
        if (idx < 16 * 27) {
          const int row = idx / 27, slot = idx - row * 27;
          s_m[wave][row][slot] = pm[e];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const uint16_t* lrow = s_lut + lk * LUT_STRIDE + 8 * g4;
      const char* mrow = reinterpret_cast<const char*>(&s_m[wave][l15][0]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint4 ent = *reinterpret_cast<const uint4*>(lrow + 32 * j);     // 8 entries: offsets 32 j + 8 g .. + 7
        const uint32_t ew[4] = {ent.x, ent.y, ent.z, ent.w};
        uint32_t byte = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint32_t e0 = ew[q] & 0xFFFFu, e1 = ew[q] >> 16;
          const uint64_t m0 = *reinterpret_cast<const uint64_t*>(mrow + (e0 >> 6));
          const uint64_t m1 = *reinterpret_cast<const uint64_t*>(mrow + (e1 >> 6));
          byte |= ((uint32_t)(m0 >> (e0 & 63)) & 1u) << (2 * q);
          byte |= ((uint32_t)(m1 >> (e1 & 63)) & 1u) << (2 * q + 1);
        }
        s_occ[wave][h * 16 + l15][4 * j + g4] = (uint8_t)byte;            // offset 32 j + 8 g + e = bit 8 g + e of word j
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4_t4*>(&s_d[wave][q * 8 + (lane >> 3)][(lane & 7) * 4]) = dv[q];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- A operand: dout^T, lane (channel 16 nt + l15, voxels 8 g4 + e), split exactly into three bf16 parts
    bf16x8_t4 dp[2][3];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      uint32_t hh[4], mm[4], ll[4];
#pragma unroll
      for (int pq = 0; pq < 4; ++pq) {
        const float x0 = s_d[wave][8 * g4 + 2 * pq][16 * nt + l15], x1 = s_d[wave][8 * g4 + 2 * pq + 1][16 * nt + l15];
        const uint32_t hp = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t4){x0, x1}, bf16x2_t4));
        const float r0 = x0 - __uint_as_float(hp << 16), r1 = x1 - __uint_as_float(hp & 0xFFFF0000u);
        const uint32_t mp = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t4){r0, r1}, bf16x2_t4));
        const float s0 = r0 - __uint_as_float(mp << 16), s1 = r1 - __uint_as_float(mp & 0xFFFF0000u);
        hh[pq] = hp; mm[pq] = mp;
        ll[pq] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t4){s0, s1}, bf16x2_t4));
      }
      dp[nt][0] = __builtin_bit_cast(bf16x8_t4, make_uint4(hh[0], hh[1], hh[2], hh[3]));
      dp[nt][1] = __builtin_bit_cast(bf16x8_t4, make_uint4(mm[0], mm[1], mm[2], mm[3]));
      dp[nt][2] = __builtin_bit_cast(bf16x8_t4, make_uint4(ll[0], ll[1], ll[2], ll[3]));
    }
    // ---- B operand per offset tile t: lane (offset 16 t + l15, voxels 8 g4 + e) = bit 16 (t & 1) + l15 of word t >> 1
    uint4 ov[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) ov[e] = *reinterpret_cast<const uint4*>(&s_occ[wave][8 * g4 + e][0]);   // (broadcast reads)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      uint32_t pk[4];
#pragma unroll
      for (int pq = 0; pq < 4; ++pq) {
        const uint4& a0 = ov[2 * pq];
        const uint4& a1 = ov[2 * pq + 1];
        const uint32_t w0 = (t >> 1) == 0 ? a0.x : ((t >> 1) == 1 ? a0.y : ((t >> 1) == 2 ? a0.z : a0.w));
        const uint32_t w1 = (t >> 1) == 0 ? a1.x : ((t >> 1) == 1 ? a1.y : ((t >> 1) == 2 ? a1.z : a1.w));
        const uint32_t b0 = (w0 >> (16 * (t & 1) + l15)) & 1u, b1 = (w1 >> (16 * (t & 1) + l15)) & 1u;
        pk[pq] = b0 * 0x3F80u + b1 * 0x3F800000u;          // bf16 1.0 in the low / high half
      }
      const bf16x8_t4 bo = __builtin_bit_cast(bf16x8_t4, make_uint4(pk[0], pk[1], pk[2], pk[3]));
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int sp = 2; sp >= 0; --sp)                    // small parts first
          acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dp[nt][sp], bo, acc[t][nt], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();                       // the next tile rewrites s_occ / s_d
  }
  // ---- workgroup partial: the four waves' accumulators summed in wave order through LDS, one partial per workgroup
  // D[M = channel 4 g4 + r (of tile nt)][N = offset l15 (of tile t)]: lane holds four consecutive channels of one offset
  __syncthreads();
  float* red = reinterpret_cast<float*>(s_lut);            // 128 x 32 floats = 16 KB (the lookup table is no longer needed)
  static_assert(sizeof(s_lut) >= 128 * 32 * 4, "reduction buffer");
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          f32x4_t4* dst = reinterpret_cast<f32x4_t4*>(red + (16 * t + l15) * 32 + 16 * nt + 4 * g4);
          *dst = (w == 0) ? acc[t][nt] : (*dst + acc[t][nt]);
        }
    }
    __syncthreads();
  }
  for (int e = tid; e < 125 * 32; e += 256) partial[(int64_t)blockIdx.x * (125 * 32) + e] = red[e];
}

int conv0_wgrad(Ctx* ctx, const float* feat, const float* dout, float* dW, float* scratch, size_t scratch_floats,
                hipStream_t stream) {
  const Plan& P = ctx->plan;
  const int64_t n0 = P.lv[0].n, size = 125 * 32;
  if (n0 == 0) {
    HIP_CHECK(hipMemsetAsync(dW, 0, (size_t)size * 4, stream));
    return EGONN_OK;
  }
  EGONN_REQUIRE(ctx->conv0_lut && P.g0 && P.t2m && P.t2s, EGONN_ERR_STATE,
                "conv0 wgrad: run the forward convolution of this plan first");
  EGONN_REQUIRE(scratch && scratch_floats >= (size_t)size, EGONN_ERR_INVALID, "conv0 wgrad: scratch too small");
  if (!feat && ctx->conv_variant != 3) {                   // unit features: the MFMA kernel (variant 3 = cross-check path)
    int64_t wgs = std::min<int64_t>(512, cdiv(cdiv(n0, 32), 4));
    wgs = std::max<int64_t>(1, std::min<int64_t>(wgs, (int64_t)(scratch_floats / (size_t)size)));
    hipLaunchKernelGGL(conv0_wgrad_unit_kernel, dim3((unsigned)wgs), dim3(256), 0, stream, dout, P.lv[0].keys, P.g0, P.t2m,
                       ctx->conv0_lut, (int32_t)n0, (int32_t)P.lv[2].n, scratch);
    launch_sum_partials(scratch, (int)wgs, size, dW, stream);
    HIP_CHECK(hipGetLastError());
    return EGONN_OK;
  }
  int64_t chunks = std::min<int64_t>(1024, cdiv(n0, 4 * C0_ROWS));
  chunks = std::max<int64_t>(1, std::min<int64_t>(chunks, (int64_t)(scratch_floats / (size_t)size)));
  const int32_t rpc = (int32_t)(cdiv(cdiv(n0, chunks), C0_ROWS) * C0_ROWS);
  chunks = cdiv(n0, rpc);
  hipLaunchKernelGGL(conv0_wgrad_kernel, dim3((unsigned)chunks), dim3(256), 0, stream, feat, dout, P.lv[0].keys, P.g0,
                     P.t2m, P.t2s, ctx->conv0_lut, (int32_t)n0, rpc, scratch);
  launch_sum_partials(scratch, (int)chunks, size, dW, stream);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------------------------------- column statistics
// mode 0: s0 = sum a            s1 = sum a^2
// mode 1: s0 = sum (a - m)^2    s1 = 0                                  (second pass of the batch variance)
// mode 2: g = a * [mask > 0] (mask nullable);  s0 = sum g,  s1 = sum g * (b - m)     (BatchNorm backward)
// mode 3: d = a - m;  s0 = sum d,  s1 = sum d^2     (single-pass batch statistics around the shift m, additive over ranks)
// block = 256 threads = (256/CP) row lanes x CP channel lanes (CP = channels padded to a power of two <= 256)
static constexpr int CS_ROWS = 512;   // rows per block
__global__ __launch_bounds__(256) void col_stats_kernel(int mode, const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ mask, const float* __restrict__ m,
                                                       int64_t n, int c, int cp, float* __restrict__ partial) {
  __shared__ float red[2][256];
  const int t = threadIdx.x;
  const int ci = t % cp, rl = t / cp, nrl = 256 / cp;
  const int64_t r0 = (int64_t)blockIdx.x * CS_ROWS, r1 = min(n, r0 + CS_ROWS);
  float s0 = 0.f, s1 = 0.f;
  if (ci < c) {
    const float mu = (mode != 0 && m) ? m[ci] : 0.f;
    if (mode == 0) {
#pragma unroll 4
      for (int64_t r = r0 + rl; r < r1; r += nrl) {
        const float av = a[r * c + ci];
        s0 += av;
        s1 = fmaf(av, av, s1);
      }
    } else if (mode == 1) {
#pragma unroll 4
      for (int64_t r = r0 + rl; r < r1; r += nrl) {
        const float d = a[r * c + ci] - mu;
        s0 = fmaf(d, d, s0);
      }
    } else if (mode == 3) {
#pragma unroll 4
      for (int64_t r = r0 + rl; r < r1; r += nrl) {
        const float d = a[r * c + ci] - mu;
        s0 += d;
        s1 = fmaf(d, d, s1);
      }
    } else if (mask) {
#pragma unroll 4
      for (int64_t r = r0 + rl; r < r1; r += nrl) {
        const float gq = (mask[r * c + ci] > 0.f) ? a[r * c + ci] : 0.f;
        s0 += gq;
        s1 = fmaf(gq, b[r * c + ci] - mu, s1);
      }
    } else {
#pragma unroll 4
      for (int64_t r = r0 + rl; r < r1; r += nrl) {
        const float gq = a[r * c + ci];
        s0 += gq;
        s1 = fmaf(gq, b[r * c + ci] - mu, s1);
      }
    }
  }
  red[0][t] = s0;
  red[1][t] = s1;
  __syncthreads();
  if (t < c) {
    float u0 = 0.f, u1 = 0.f;
    for (int k = 0; k < nrl; ++k) {
      u0 += red[0][k * cp + t];
      u1 += red[1][k * cp + t];
    }
    partial[((int64_t)blockIdx.x * 2 + 0) * c + t] = u0;
    partial[((int64_t)blockIdx.x * 2 + 1) * c + t] = u1;
  }
}


// The same statistics with 16-byte accesses (channel counts that are multiples of 4: every BatchNorm of the models): thread =
// (row lane, four consecutive channels), 1024 rows per block, four rows in flight per thread.  The scalar kernel above ran at
// a third of the HBM rate (36 us per call on average over the 50 calls of a 32-scan step, 1.8 ms per step).
static constexpr int CS4_ROWS = 1024;   // rows per block on big maps; small maps get fewer (a block of a 1 500-row map walked
                                        // 128 rows per thread one after the other: 28 us forward, 60 us backward per call)
__global__ __launch_bounds__(256) void col_stats4_kernel(int mode, const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ mask, const float* __restrict__ m,
                                                        int64_t n, int c, int rows_per_block, float* __restrict__ partial) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ f4 red[2][256];
  const int t = threadIdx.x;
  const int lq = c >> 2;                                   // lanes per row (8 .. 64), a power of two or 24 / 48
  const int nrl = 256 / lq;                                // row lanes (threads beyond nrl * lq idle)
  const int ci = t % lq, rl = t / lq;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(n, r0 + rows_per_block);
  f4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  if (rl < nrl) {
    f4 mu = {0.f, 0.f, 0.f, 0.f};
    if (mode != 0 && m) mu = *reinterpret_cast<const f4*>(m + 4 * ci);
    const f4* a4 = reinterpret_cast<const f4*>(a) + ci;
    const f4* b4 = reinterpret_cast<const f4*>(b) + ci;
    const f4* k4 = reinterpret_cast<const f4*>(mask) + ci;
    for (int64_t r = r0 + rl; r < r1; r += 4 * nrl) {
      f4 av[4], bv[4], kv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t rr = r + (int64_t)u * nrl;
        const bool ok = rr < r1;
        av[u] = ok ? a4[rr * lq] : (f4){0.f, 0.f, 0.f, 0.f};
        if (mode == 2) {
          bv[u] = ok ? b4[rr * lq] : mu;
          if (mask) kv[u] = ok ? k4[rr * lq] : (f4){0.f, 0.f, 0.f, 0.f};
        }
        if ((mode == 1 || mode == 3) && !ok) av[u] = mu;   // d = 0 for the rows past the end
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (mode == 0) {
          s0 += av[u];
          s1 += av[u] * av[u];
        } else if (mode == 1) {
          const f4 d = av[u] - mu;
          s0 += d * d;
        } else if (mode == 3) {
          const f4 d = av[u] - mu;
          s0 += d;
          s1 += d * d;
        } else {
          f4 g = av[u];
          if (mask) {
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = kv[u][q] > 0.f ? g[q] : 0.f;
          }
          s0 += g;
          s1 += g * (bv[u] - mu);
        }
      }
    }
  }
  red[0][t] = s0;
  red[1][t] = s1;
  __syncthreads();
  if (t < lq) {
    f4 u0 = {0.f, 0.f, 0.f, 0.f}, u1 = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < nrl; ++k) {
      u0 += red[0][k * lq + t];
      u1 += red[1][k * lq + t];
    }
    *reinterpret_cast<f4*>(partial + ((int64_t)blockIdx.x * 2 + 0) * c + 4 * t) = u0;
    *reinterpret_cast<f4*>(partial + ((int64_t)blockIdx.x * 2 + 1) * c + 4 * t) = u1;
  }
}

// out[i] = sum over chunks of partial[ch][i]: one wave per output value (few values, many chunks), fp64, fixed order
__global__ __launch_bounds__(64) void sum_partials_wave_kernel(const float* __restrict__ partial, int chunks, int64_t size,
                                                              float* __restrict__ out) {
  const int64_t i = blockIdx.x;
  double s = 0.0;
  for (int ch = threadIdx.x; ch < chunks; ch += 64) s += (double)partial[(int64_t)ch * size + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (threadIdx.x == 0) out[i] = (float)s;
}

// the same with a whole workgroup per output value (the BatchNorm statistics: 64-512 outputs, 500-1 600 partial rows): thread t
// adds rows t, t + 256, ... in ascending order, the 256 sums are added in a fixed binary tree
__global__ __launch_bounds__(256) void sum_partials_block_kernel(const float* __restrict__ partial, int chunks, int64_t size,
                                                                float* __restrict__ out) {
  __shared__ double red[256];
  const int64_t i = blockIdx.x;
  const int t = threadIdx.x;
  double s = 0.0;
  for (int ch = t; ch < chunks; ch += 256) s += (double)partial[(int64_t)ch * size + i];
  red[t] = s;
  __syncthreads();
#pragma unroll
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  if (t == 0) out[i] = (float)red[0];
}

int col_stats(int mode, const float* a, const float* b, const float* mask, const float* m, int64_t n, int c, float* out2c,
              float* scratch, size_t scratch_floats, hipStream_t stream) {
  EGONN_REQUIRE(c >= 1 && c <= 256, EGONN_ERR_INVALID, "col_stats: %d channels unsupported (1..256)", c);
  EGONN_REQUIRE(mode >= 0 && mode <= 3 && a && (mode != 2 || b), EGONN_ERR_INVALID, "col_stats: bad arguments");
  if (n == 0) {
    HIP_CHECK(hipMemsetAsync(out2c, 0, (size_t)2 * c * 4, stream));
    return EGONN_OK;
  }
  const bool aligned16 = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(mask) |
                           reinterpret_cast<uintptr_t>(m)) & 15u) == 0;
  if (c % 4 == 0 && c >= 32 && 256 % (c / 4) == 0 && aligned16) {      // 32 / 64 / 128 / 256 channels: 16-byte accesses
    // >= ~500 blocks whenever the map has the rows for it (a function of n only: deterministic)
    int rpb = CS4_ROWS;
    while (rpb > 32 && cdiv(n, rpb) < 512) rpb >>= 1;
    // never more than 1 023 blocks below CS4_ROWS rows per block, so 2*c*max(1024, ceil(n/512)) floats (the header's contract)
    // always hold them; a caller with less gets coarser blocks (another fixed summation order) instead of an error
    while (rpb < CS4_ROWS && (size_t)cdiv(n, rpb) * 2 * c > scratch_floats) rpb <<= 1;
    const int64_t blocks4 = cdiv(n, rpb);
    EGONN_REQUIRE(scratch && scratch_floats >= (size_t)blocks4 * 2 * c, EGONN_ERR_INVALID,
                  "col_stats: scratch too small (%zu < %lld floats)", scratch_floats, (long long)(blocks4 * 2 * c));
    hipLaunchKernelGGL(col_stats4_kernel, dim3((unsigned)blocks4), dim3(256), 0, stream, mode, a, b, mask, m, n, c, rpb, scratch);
    if (blocks4 >= 256)
      hipLaunchKernelGGL(sum_partials_block_kernel, dim3((unsigned)(2 * c)), dim3(256), 0, stream, scratch, (int)blocks4,
                         (int64_t)2 * c, out2c);
    else
      hipLaunchKernelGGL(sum_partials_wave_kernel, dim3((unsigned)(2 * c)), dim3(64), 0, stream, scratch, (int)blocks4,
                         (int64_t)2 * c, out2c);
    HIP_CHECK(hipGetLastError());
    return EGONN_OK;
  }
  int cp = 1;
  while (cp < c) cp <<= 1;
  const int64_t blocks = cdiv(n, CS_ROWS);
  EGONN_REQUIRE(scratch && scratch_floats >= (size_t)blocks * 2 * c, EGONN_ERR_INVALID,
                "col_stats: scratch too small (%zu < %lld floats)", scratch_floats, (long long)(blocks * 2 * c));
  hipLaunchKernelGGL(col_stats_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, mode, a, b, mask, m, n, c, cp,
                     scratch);
  hipLaunchKernelGGL(sum_partials_wave_kernel, dim3((unsigned)(2 * c)), dim3(64), 0, stream, scratch, (int)blocks,
                     (int64_t)2 * c, out2c);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------------------------------- BatchNorm vector math
// forward: sums (2,c) = [sum d, sum d^2] around shift m (whole batch, after the SyncBN all-reduce), count n ->
//   mean, invstd, scale = w * invstd, shift = b - mean * scale (out4: 4 x c), running statistics updated in place
//   (momentum; unbiased variance), exactly nn.BatchNorm1d's bookkeeping.
__global__ void bn_fwd_finalize_kernel(const float* __restrict__ sums, const float* __restrict__ m, double n, int c,
                                       const float* __restrict__ w, const float* __restrict__ b, float eps, float momentum,
                                       float* __restrict__ running_mean, float* __restrict__ running_var,
                                       float* __restrict__ out4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  const double sd = sums[i], sq = sums[c + i];
  const double dm = sd / n;
  const double mean = (double)m[i] + dm;
  double var = sq / n - dm * dm;
  var = var > 0.0 ? var : 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float scale = w[i] * invstd;
  out4[i] = (float)mean;
  out4[c + i] = invstd;
  out4[2 * c + i] = scale;
  out4[3 * c + i] = b[i] - (float)mean * scale;
  if (running_mean) {
    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
    running_mean[i] = (1.f - momentum) * running_mean[i] + momentum * (float)mean;
    running_var[i] = (1.f - momentum) * running_var[i] + momentum * (float)unbiased;
  }
}
// backward: local sums (2,c) = [sum g', sum g'(x - mean)] of this rank, global sums (after the all-reduce), count n ->
//   out5: A = w invstd, B = -w invstd^3 S2/n, C = -B mean - A S1/n, dgamma = S2_local invstd, dbeta = S1_local
__global__ void bn_bwd_finalize_kernel(const float* __restrict__ local, const float* __restrict__ global, double n, int c,
                                       const float* __restrict__ w, const float* __restrict__ mean,
                                       const float* __restrict__ invstd, float* __restrict__ out5) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  const double inv = invstd[i], wi = w[i];
  const double A = wi * inv;
  const double Bc = -wi * inv * inv * inv * (double)global[c + i] / n;
  const double Cc = -Bc * (double)mean[i] - A * (double)global[i] / n;
  out5[i] = (float)A;
  out5[c + i] = (float)Bc;
  out5[2 * c + i] = (float)Cc;
  out5[3 * c + i] = (float)((double)local[c + i] * inv);
  out5[4 * c + i] = local[i];
}
int bn_fwd_finalize(const float* sums, const float* m, double n, int c, const float* w, const float* b, float eps,
                    float momentum, float* running_mean, float* running_var, float* out4, hipStream_t stream) {
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((unsigned)cdiv(c, 64)), dim3(64), 0, stream, sums, m, n, c, w, b, eps,
                     momentum, running_mean, running_var, out4);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}
int bn_bwd_finalize(const float* local, const float* global, double n, int c, const float* w, const float* mean,
                    const float* invstd, float* out5, hipStream_t stream) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)cdiv(c, 64)), dim3(64), 0, stream, local, global, n, c, w, mean,
                     invstd, out5);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------------------------------- element-wise passes
// out = act(x * A[c] + B[c])
__global__ void affine_act_kernel(const float* __restrict__ x, const float* __restrict__ A, const float* __restrict__ Bv,
                                  int64_t total, int c, int relu, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ci = (int)(i % c);
  float v = fmaf(x[i], A[ci], Bv[ci]);
  if (relu) v = fmaxf(v, 0.f);
  out[i] = v;
}
// out = A[c] * (g * [mask > 0]) + B[c] * x + C[c]      (BatchNorm backward; mask nullable)
__global__ void affine3_kernel(const float* __restrict__ g, const float* __restrict__ mask, const float* __restrict__ x,
                               const float* __restrict__ A, const float* __restrict__ Bv, const float* __restrict__ Cv,
                               int64_t total, int c, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ci = (int)(i % c);
  const float gq = (mask && !(mask[i] > 0.f)) ? 0.f : g[i];
  out[i] = fmaf(A[ci], gq, fmaf(Bv[ci], x[i], Cv[ci]));
}
int affine_act(const float* x, const float* A, const float* B, int64_t n, int c, int relu, float* out, hipStream_t stream) {
  const int64_t total = n * c;
  if (total == 0) return EGONN_OK;
  hipLaunchKernelGGL(affine_act_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, x, A, B, total, c, relu, out);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}
int affine3(const float* g, const float* mask, const float* x, const float* A, const float* B, const float* C, int64_t n,
            int c, float* out, hipStream_t stream) {
  const int64_t total = n * c;
  if (total == 0) return EGONN_OK;
  hipLaunchKernelGGL(affine3_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, g, mask, x, A, B, C, total, c,
                     out);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

__device__ static inline int sample_of(const int32_t* __restrict__ boff, int B, int32_t r) {
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (boff[mid] <= r) lo = mid; else hi = mid;
  }
  return lo;
}

// out = relu(x * gate[b] + res)   (gate nullable -> 1; res nullable -> 0)
__global__ void gate_res_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gate, const float* __restrict__ res,
                                    const int32_t* __restrict__ boff, int B, int64_t total, int c, int relu,
                                    float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int32_t r = (int32_t)(i / c);
  const int ci = (int)(i - (int64_t)r * c);
  float v = x[i];
  if (gate) v *= gate[(int64_t)sample_of(boff, B, r) * c + ci];
  if (res) v += res[i];
  out[i] = relu ? fmaxf(v, 0.f) : v;
}
// d = dout * [out > 0];  dres = d;  dx = d * gate[b]
__global__ void gate_res_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                    const float* __restrict__ gate, const int32_t* __restrict__ boff, int B, int64_t total,
                                    int c, float* __restrict__ dx, float* __restrict__ dres) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int32_t r = (int32_t)(i / c);
  const int ci = (int)(i - (int64_t)r * c);
  const float d = (out && !(out[i] > 0.f)) ? 0.f : dout[i];
  if (dres) dres[i] = d;
  dx[i] = gate ? d * gate[(int64_t)sample_of(boff, B, r) * c + ci] : d;
}
// out[r][c] = v[b(r)][c] * (mean ? 1/n_b : 1)     (backward of the per-sample average pooling / broadcast)
__global__ void seg_broadcast_kernel(const float* __restrict__ v, const int32_t* __restrict__ boff, int B, int64_t total,
                                     int c, int mean, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int32_t r = (int32_t)(i / c);
  const int ci = (int)(i - (int64_t)r * c);
  const int b = sample_of(boff, B, r);
  float s = v[(int64_t)b * c + ci];
  if (mean) s /= (float)(boff[b + 1] - boff[b]);
  out[i] = s;
}

// per-sample column sums of f(a, b): partial[b][chunk][c], same two-stage layout as segment_partial_sums
//   mode 0: a * b            mode 1: t^p * ln t, t = max(a, 1e-6)        mode 2: (a * [b > 0]) * x2   (x2 = third input)
__global__ __launch_bounds__(256) void seg_sums2_kernel(int mode, const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ x2, const float* __restrict__ pexp,
                                                       const int32_t* __restrict__ boff, int c, float* __restrict__ partial) {
  __shared__ float red[256];
  const int bi = blockIdx.y, ch = blockIdx.x;
  const int32_t s = boff[bi], e = boff[bi + 1];
  const int64_t len = e - s;
  const int32_t r0 = s + (int32_t)(len * ch / SEG_CHUNKS), r1 = s + (int32_t)(len * (ch + 1) / SEG_CHUNKS);
  const int tid = threadIdx.x;
  const int rl = tid / c, cidx = tid - rl * c, nrl = 256 / c;
  const float p = pexp ? pexp[0] : 1.f;
  float acc = 0.f;
  if (rl < nrl) {
    for (int32_t r = r0 + rl; r < r1; r += nrl) {
      const int64_t i = (int64_t)r * c + cidx;
      float v;
      if (mode == 0) v = a[i] * b[i];
      else if (mode == 1) {
        const float tq = fmaxf(a[i], 1e-6f);
        v = powf(tq, p) * logf(tq);
      } else v = ((b[i] > 0.f) ? a[i] : 0.f) * x2[i];
      acc += v;
    }
  }
  red[tid] = acc;
  __syncthreads();
  if (tid < c) {
    float sum = 0.f;
    for (int k = 0; k < nrl; ++k) sum += red[k * c + tid];
    partial[((int64_t)bi * SEG_CHUNKS + ch) * c + tid] = sum;
  }
}
__global__ void seg_finish_kernel(const float* __restrict__ partial, int64_t bc, int c, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bc) return;
  const int64_t b = i / c, ci = i % c;
  float s = 0.f;
  for (int ch = 0; ch < SEG_CHUNKS; ++ch) s += partial[(b * SEG_CHUNKS + ch) * c + ci];
  out[i] = s;
}
int seg_sums2(int mode, const float* a, const float* b, const float* x2, const float* p, const int32_t* boff, int B, int c,
              float* out_bc, float* scratch, size_t scratch_floats, hipStream_t stream) {
  EGONN_REQUIRE(c >= 1 && c <= 256 && 256 % c == 0, EGONN_ERR_INVALID, "segment sums: %d channels unsupported", c);
  EGONN_REQUIRE(scratch && scratch_floats >= (size_t)B * SEG_CHUNKS * c, EGONN_ERR_INVALID, "segment sums: scratch too small");
  hipLaunchKernelGGL(seg_sums2_kernel, dim3(SEG_CHUNKS, B), dim3(256), 0, stream, mode, a, b, x2, p, boff, c, scratch);
  hipLaunchKernelGGL(seg_finish_kernel, dim3((unsigned)cdiv((int64_t)B * c, 256)), dim3(256), 0, stream, scratch,
                     (int64_t)B * c, c, out_bc);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

int gate_residual_forward(const float* x, const float* gate, const float* res, const int32_t* boff, int B, int64_t n, int c,
                          int relu, float* out, hipStream_t stream) {
  const int64_t total = n * c;
  if (total == 0) return EGONN_OK;
  hipLaunchKernelGGL(gate_res_fwd_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, x, gate, res, boff, B,
                     total, c, relu, out);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}
int gate_residual_backward(const float* dout, const float* out, const float* gate, const int32_t* boff, int B, int64_t n,
                           int c, float* dx, float* dres, hipStream_t stream) {
  const int64_t total = n * c;
  if (total == 0) return EGONN_OK;
  hipLaunchKernelGGL(gate_res_bwd_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, dout, out, gate, boff, B,
                     total, c, dx, dres);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}
int seg_broadcast(const float* v, const int32_t* boff, int B, int64_t n, int c, int mean, float* out, hipStream_t stream) {
  const int64_t total = n * c;
  if (total == 0) return EGONN_OK;
  hipLaunchKernelGGL(seg_broadcast_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, v, boff, B, total, c,
                     mean, out);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// GeM backward w.r.t. the rows: dx[r][c] = coef[b][c] * t^(p-1) * [x >= 1e-6], t = max(x, 1e-6)
// (coef = dout * out^(1-p) / n_b is formed on the host side from (B, C) tensors)
__global__ void gem_bwd_kernel(const float* __restrict__ x, const float* __restrict__ coef, const float* __restrict__ pexp,
                               const int32_t* __restrict__ boff, int B, int64_t total, int c, float* __restrict__ dx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int32_t r = (int32_t)(i / c);
  const int ci = (int)(i - (int64_t)r * c);
  const float v = x[i];
  float o = 0.f;
  if (v >= 1e-6f) o = coef[(int64_t)sample_of(boff, B, r) * c + ci] * powf(v, pexp[0] - 1.f);
  dx[i] = o;
}
int gem_backward_rows(const float* x, const float* coef, const float* p, const int32_t* boff, int B, int64_t n, int c,
                      float* dx, hipStream_t stream) {
  const int64_t total = n * c;
  if (total == 0) return EGONN_OK;
  hipLaunchKernelGGL(gem_bwd_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, x, coef, p, boff, B, total, c,
                     dx);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------------------------------ ECA gate on (B, C) means
// gate = sigmoid(Conv1d_k(mean)) over the channel axis, zero padding (k-1)/2, no bias (layers/eca_block.py:17-19,28-31).
// One workgroup per sample forward; backward: dz = dgate * g * (1 - g), dmean = correlation of dz with w,
// dw[j] = sum_{b,c} dz[b][c] * mean[b][c + j - pad] reduced by ONE workgroup in fixed order (deterministic).
__global__ void eca_gate_fwd_kernel(const float* __restrict__ mean, const float* __restrict__ w, int ks, int c,
                                    float* __restrict__ gate) {
  const int b = blockIdx.x, t = threadIdx.x;
  if (t >= c) return;
  const int pad = (ks - 1) / 2;
  float z = 0.f;
  for (int j = 0; j < ks; ++j) {
    const int q = t + j - pad;
    if (q >= 0 && q < c) z = fmaf(w[j], mean[(int64_t)b * c + q], z);
  }
  gate[(int64_t)b * c + t] = 1.f / (1.f + expf(-z));
}
// One workgroup (the (B, C) gate is 4 096 values at batch 32).  dz = dgate * g (1 - g) and the pooled means are staged once in
// LDS (the first version read gate / dgate / mean from global memory inside every one of its 6 passes: 38 us of dependent
// loads per call); sums in the same order as before (thread-strided, then a binary tree): bitwise the same results.
static constexpr int ECA_BWD_LDS = 8192;                  // (B * C) values the staged path holds
__global__ __launch_bounds__(256) void eca_gate_bwd_kernel(const float* __restrict__ dgate, const float* __restrict__ gate,
                                                          const float* __restrict__ mean, const float* __restrict__ w,
                                                          int ks, int B, int c, float* __restrict__ dmean,
                                                          float* __restrict__ dw) {
  __shared__ float red[256];
  __shared__ float s_dz[ECA_BWD_LDS];
  __shared__ float s_mean[ECA_BWD_LDS];
  __shared__ float s_w[16];
  const int t = threadIdx.x;
  const int pad = (ks - 1) / 2;
  const int total = B * c;
  const bool staged = total <= ECA_BWD_LDS;
  if (staged) {
    for (int e = t; e < total; e += 256) {
      const float g = gate[e];
      s_dz[e] = dgate[e] * g * (1.f - g);
      s_mean[e] = mean[e];
    }
    if (t < ks) s_w[t] = w[t];
    __syncthreads();
  }
  auto dz_at = [&](int e) {
    if (staged) return s_dz[e];
    const float g = gate[e];
    return dgate[e] * g * (1.f - g);
  };
  auto mean_at = [&](int e) { return staged ? s_mean[e] : mean[e]; };
  auto w_at = [&](int j) { return staged ? s_w[j] : w[j]; };
  for (int e = t; e < total; e += 256) {                 // dmean[b][q] = sum_j w[j] * dz[b][q - j + pad]
    const int b = e / c, q = e - b * c;
    float s = 0.f;
    for (int j = 0; j < ks; ++j) {
      const int cc = q - j + pad;
      if (cc >= 0 && cc < c) s = fmaf(w_at(j), dz_at(b * c + cc), s);
    }
    dmean[e] = s;
  }
  for (int j = 0; j < ks; ++j) {
    float s = 0.f;
    for (int e = t; e < total; e += 256) {
      const int b = e / c, cc = e - b * c;
      const int q = cc + j - pad;
      if (q >= 0 && q < c) s = fmaf(dz_at(e), mean_at(b * c + q), s);
    }
    red[t] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (t < o) red[t] += red[t + o];
      __syncthreads();
    }
    if (t == 0) dw[j] = red[0];
    __syncthreads();
  }
}
int eca_gate_forward(const float* mean, const float* w, int ks, int B, int c, float* gate, hipStream_t stream) {
  EGONN_REQUIRE(c >= 1 && c <= 256 && ks >= 1 && ks <= 15 && (ks & 1), EGONN_ERR_INVALID, "eca_gate: bad shape");
  if (B == 0) return EGONN_OK;
  hipLaunchKernelGGL(eca_gate_fwd_kernel, dim3((unsigned)B), dim3(256), 0, stream, mean, w, ks, c, gate);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}
int eca_gate_backward(const float* dgate, const float* gate, const float* mean, const float* w, int ks, int B, int c,
                      float* dmean, float* dw, hipStream_t stream) {
  EGONN_REQUIRE(c >= 1 && c <= 256 && ks >= 1 && ks <= 15 && (ks & 1), EGONN_ERR_INVALID, "eca_gate: bad shape");
  hipLaunchKernelGGL(eca_gate_bwd_kernel, dim3(1), dim3(256), 0, stream, dgate, gate, mean, w, ks, B, c, dmean, dw);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------------------------------ activations / L2 normalisation
// grad_in = grad_out * act'(.) expressed through the activation's OUTPUT y:
//   relu: [y > 0]   tanh: 1 - y^2   softplus: 1 - exp(-y)  (= sigmoid(x))   sigmoid: y (1 - y)
__global__ void act_bwd_kernel(int act, const float* __restrict__ g, const float* __restrict__ y, int64_t total,
                               float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float yv = y[i];
  float d = 1.f;
  if (act == ACT_RELU) d = yv > 0.f ? 1.f : 0.f;
  else if (act == ACT_TANH) d = 1.f - yv * yv;
  else if (act == ACT_SOFTPLUS) d = 1.f - expf(-yv);
  else if (act == ACT_SIGMOID) d = yv * (1.f - yv);
  out[i] = g[i] * d;
}
int act_backward(int act, const float* g, const float* y, int64_t n, int c, float* out, hipStream_t stream) {
  const int64_t total = n * c;
  if (total == 0) return EGONN_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, act, g, y, total, out);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// F.normalize(x, p=2, dim=1, eps=1e-12): y = x / max(|x|, eps).  One wave per row.
// backward (|x| > eps): dx = (g - y (g . y)) / |x| ;  (|x| <= eps): dx = g / eps
__global__ __launch_bounds__(256) void l2norm_fwd_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                            int64_t n, int c, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float* p = x + row * c;
  float ss = 0.f, gy = 0.f;
  for (int i = lane; i < c; i += 64) ss = fmaf(p[i], p[i], ss);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float nrm = sqrtf(ss);
  const float inv = 1.f / fmaxf(nrm, 1e-12f);
  if (!g) {
    for (int i = lane; i < c; i += 64) out[row * c + i] = p[i] * inv;
    return;
  }
  const float* q = g + row * c;
  for (int i = lane; i < c; i += 64) gy = fmaf(q[i], p[i] * inv, gy);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gy += __shfl_xor(gy, o, 64);
  const bool clamped = !(nrm > 1e-12f);
  for (int i = lane; i < c; i += 64) out[row * c + i] = clamped ? q[i] * inv : (q[i] - p[i] * inv * gy) * inv;
}
int l2norm_rows(const float* x, const float* g, int64_t n, int c, float* out, hipStream_t stream) {
  if (n == 0) return EGONN_OK;
  hipLaunchKernelGGL(l2norm_fwd_bwd_kernel, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, stream, x, g, n, c, out);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

}  // namespace egonn
