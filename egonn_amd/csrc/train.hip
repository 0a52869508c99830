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
                                                        int32_t rows_per_chunk, float* __restrict__ partial) {
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
  const int32_t r0 = chunk * rows_per_chunk;
  const int32_t r1 = (int32_t)min((int64_t)n_out, (int64_t)r0 + rows_per_chunk);
  for (int32_t base = r0; base < r1; base += 256) {
    const int32_t o = base + t;
    int32_t j = -1;
    if (o < r1) j = nbr ? nbr[(int64_t)o * K + k] : o;
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
    for (int p0 = 0; p0 < cnt; p0 += PB) {
      for (int e = t; e < PB * (CIN / 4); e += 256) {
        const int pr = e / (CIN / 4), c4 = e % (CIN / 4);
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p0 + pr < cnt) val = *reinterpret_cast<const float4*>(in + (int64_t)s_pair[p0 + pr].x * CIN + c4 * 4);
        *reinterpret_cast<float4*>(&s_a[pr * LDA + c4 * 4]) = val;
      }
      for (int e = t; e < PB * (COUT / 4); e += 256) {
        const int pr = e / (COUT / 4), c4 = e % (COUT / 4);
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p0 + pr < cnt) val = *reinterpret_cast<const float4*>(dout + (int64_t)s_pair[p0 + pr].y * COUT + c4 * 4);
        *reinterpret_cast<float4*>(&s_b[pr * LDB + c4 * 4]) = val;
      }
      __syncthreads();
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

int conv_wgrad(const float* in, const float* dout, const int32_t* nbr, int64_t n_out, int K, int cin, int cout,
               float* dW, float* scratch, size_t scratch_floats, hipStream_t stream) {
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
    int64_t chunks = std::max<int64_t>(1, 2048 / (int64_t)K);
    chunks = std::min<int64_t>(chunks, cdiv(n_out, 512));
    chunks = std::max<int64_t>(1, std::min<int64_t>(chunks, (int64_t)(scratch_floats / (size_t)size)));
    const int32_t rpc = (int32_t)cdiv(n_out, chunks);
    chunks = cdiv(n_out, rpc);
    const dim3 grid((unsigned)chunks, (unsigned)K);
#define EGONN_WGRAD_CASE(CI, CO)                                                                                 \
  if (cin == CI && cout == CO)                                                                                   \
    hipLaunchKernelGGL((wgrad_mfma_kernel<CI, CO>), grid, dim3(256), 0, stream, in, dout, nbr, (int32_t)n_out, K, rpc, \
                       scratch);
    EGONN_WGRAD_CASE(32, 32)
    EGONN_WGRAD_CASE(32, 64)
    EGONN_WGRAD_CASE(64, 64)
    EGONN_WGRAD_CASE(64, 128)
    EGONN_WGRAD_CASE(128, 128)
#undef EGONN_WGRAD_CASE
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)cdiv(size, 256)), dim3(256), 0, stream, scratch, (int)chunks,
                       size, dW);
    HIP_CHECK(hipGetLastError());
    return EGONN_OK;
  }
  const bool small = cin <= 32 && cout <= 32;
  const int T = small ? 32 : 64, NT = small ? 64 : 256;
  const int tiles = (int)(cdiv(cin, T) * cdiv(cout, T));
  int64_t chunks = std::max<int64_t>(1, 4096 / ((int64_t)K * tiles));
  chunks = std::min<int64_t>(chunks, cdiv(n_out, 2 * NT));
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
  hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)cdiv(size, 256)), dim3(256), 0, stream, scratch, (int)chunks, size,
                     dW);
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
  int64_t chunks = std::min<int64_t>(1024, cdiv(n0, 4 * C0_ROWS));
  chunks = std::max<int64_t>(1, std::min<int64_t>(chunks, (int64_t)(scratch_floats / (size_t)size)));
  EGONN_REQUIRE(scratch && scratch_floats >= (size_t)size, EGONN_ERR_INVALID, "conv0 wgrad: scratch too small");
  const int32_t rpc = (int32_t)(cdiv(cdiv(n0, chunks), C0_ROWS) * C0_ROWS);
  chunks = cdiv(n0, rpc);
  hipLaunchKernelGGL(conv0_wgrad_kernel, dim3((unsigned)chunks), dim3(256), 0, stream, feat, dout, P.lv[0].keys, P.g0,
                     P.t2m, P.t2s, ctx->conv0_lut, (int32_t)n0, rpc, scratch);
  hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)cdiv(size, 256)), dim3(256), 0, stream, scratch, (int)chunks, size,
                     dW);
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

int col_stats(int mode, const float* a, const float* b, const float* mask, const float* m, int64_t n, int c, float* out2c,
              float* scratch, size_t scratch_floats, hipStream_t stream) {
  EGONN_REQUIRE(c >= 1 && c <= 256, EGONN_ERR_INVALID, "col_stats: %d channels unsupported (1..256)", c);
  EGONN_REQUIRE(mode >= 0 && mode <= 3 && a && (mode != 2 || b), EGONN_ERR_INVALID, "col_stats: bad arguments");
  if (n == 0) {
    HIP_CHECK(hipMemsetAsync(out2c, 0, (size_t)2 * c * 4, stream));
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
__global__ __launch_bounds__(256) void eca_gate_bwd_kernel(const float* __restrict__ dgate, const float* __restrict__ gate,
                                                          const float* __restrict__ mean, const float* __restrict__ w,
                                                          int ks, int B, int c, float* __restrict__ dmean,
                                                          float* __restrict__ dw) {
  __shared__ float red[256];
  const int t = threadIdx.x;
  const int pad = (ks - 1) / 2;
  const int total = B * c;
  for (int e = t; e < total; e += 256) {                 // dmean[b][q] = sum_j w[j] * dz[b][q - j + pad]
    const int b = e / c, q = e - b * c;
    float s = 0.f;
    for (int j = 0; j < ks; ++j) {
      const int cc = q - j + pad;
      if (cc >= 0 && cc < c) {
        const float g = gate[(int64_t)b * c + cc];
        s = fmaf(w[j], dgate[(int64_t)b * c + cc] * g * (1.f - g), s);
      }
    }
    dmean[e] = s;
  }
  for (int j = 0; j < ks; ++j) {
    float s = 0.f;
    for (int e = t; e < total; e += 256) {
      const int b = e / c, cc = e - b * c;
      const int q = cc + j - pad;
      if (q >= 0 && q < c) {
        const float g = gate[e];
        s = fmaf(dgate[e] * g * (1.f - g), mean[(int64_t)b * c + q], s);
      }
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
