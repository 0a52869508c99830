// Row-wise (dense) kernels of the EgoNN path: 1x1 convolutions / nn.Linear on exact-f32 MFMA, folded
// BatchNorm, ECA gate, GeM pooling, L2 normalisation, keypoint positions and top-k keypoint selection.
//
// Reference call sites: MinkowskiConvolution(k=1) models/minkgl.py:43,124 ; MinkowskiLinear
// models/minkgl.py:167-217 ; MinkowskiBatchNorm (nn.BatchNorm1d eval) ; ECALayer layers/eca_block.py:11-36 ;
// GeM layers/pooling.py:82-86 ; MinkowskiFunctional.normalize models/minkgl.py:223 ;
// Quantizer.keypoint_position datasets/quantization.py:60-72,93-103 ; torch.topk eval/evaluate.py:359.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace egonn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ static inline float apply_act(float v, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.f);
    case ACT_TANH: return tanhf(v);
    case ACT_SOFTPLUS: return v > 20.f ? v : log1pf(expf(v));     // torch softplus(beta=1, threshold=20)
    case ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}

// ------------------------------------------------------------------ dense GEMM + epilogue
// Workgroup = 64 rows x 64 output columns (grid.y walks the column groups), wave = 16 rows x 64 columns as two
// passes of two 16-wide v_mfma_f32_16x16x4_f32 column tiles.  The wave's A fragment (16 rows x CIN) is loaded
// once into registers; all B loads of a pass are issued before its MFMA chains (CIN is a template parameter so
// the loops unroll and the loads batch).  The contraction index inside a 16-wide k-block is permuted (lane
// group g owns ci = 16t + 4g .. 4g+3) so that A and (out,in)-layout B come in as one float4 per lane.
// IN_BF16: the input rows are bf16 (configs[2] feature maps; widened exactly to fp32 in registers — the weights and the
// arithmetic stay fp32); io bit 0: residual is bf16, bit 1: out is bf16.
__device__ static inline float bf2f(uint32_t h) { return __uint_as_float(h << 16); }
__device__ static inline uint32_t f2bf_rn(float a) {
  uint32_t u = __float_as_uint(a);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}
template <int W_OUT_IN, int CIN, bool IN_BF16>
__global__ __launch_bounds__(256) void dense_kernel(const void* __restrict__ in_v, int64_t n,
                                                    const float* __restrict__ W, int cout,
                                                    const float* __restrict__ bias, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, int act,
                                                    const void* __restrict__ residual_v, void* __restrict__ out_v, int io,
                                                    const int32_t* __restrict__ n_dev) {
  if (n_dev) n = min((int64_t)*n_dev, n);               // reserved plans: the grid is sized for the capacity
  const float* in = reinterpret_cast<const float*>(in_v);
  const float* residual = reinterpret_cast<const float*>(residual_v);
  float* out = reinterpret_cast<float*>(out_v);
  constexpr int KS = CIN / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, g4 = lane >> 4;
  const int64_t row_base = ((int64_t)blockIdx.x * 4 + wave) * 16;
  if (row_base >= n) return;
  const int64_t row = row_base + l15;
  float4 a[KS];
#pragma unroll
  for (int t = 0; t < KS; ++t) {
    a[t] = make_float4(0, 0, 0, 0);
    if (row < n) {
      if constexpr (IN_BF16) {
        const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(in_v) + row * CIN + 16 * t + 4 * g4);
        a[t] = make_float4(bf2f(h.x & 0xFFFFu), bf2f(h.x >> 16), bf2f(h.y & 0xFFFFu), bf2f(h.y >> 16));
      } else {
        a[t] = *reinterpret_cast<const float4*>(in + row * CIN + 16 * t + 4 * g4);
      }
    }
  }
  const int ncol0 = blockIdx.y * 64;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    const int n0 = ncol0 + pass * 32;
    if (n0 >= cout) break;
    float4 b[2][KS];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col = n0 + nt * 16 + l15;
#pragma unroll
      for (int t = 0; t < KS; ++t) {
        float4 v = make_float4(0, 0, 0, 0);
        if (col < cout) {
          if (W_OUT_IN) {
            v = *reinterpret_cast<const float4*>(W + (int64_t)col * CIN + 16 * t + 4 * g4);
          } else {
            const float* wp = W + (int64_t)(16 * t + 4 * g4) * cout + col;
            v.x = wp[0];
            v.y = wp[cout];
            v.z = wp[2 * (int64_t)cout];
            v.w = wp[3 * (int64_t)cout];
          }
        }
        b[nt][t] = v;
      }
    }
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int t = 0; t < KS; ++t) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].x, b[nt][t].x, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].y, b[nt][t].y, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].z, b[nt][t].z, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].w, b[nt][t].w, acc[nt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int col = n0 + nt * 16 + l15;
      if (col < cout) {
        const float bi = bias ? bias[col] : 0.f;
        const float sc = scale ? scale[col] : 1.f, sh = scale ? shift[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t orow = row_base + 4 * g4 + r;
          if (orow < n) {
            float v = acc[nt][r] + bi;
            if (scale) v = v * sc + sh;
            v = apply_act(v, act);
            if (residual_v)
              v += (io & 1) ? bf2f(reinterpret_cast<const uint16_t*>(residual_v)[orow * cout + col]) : residual[orow * cout + col];
            if (io & 2) reinterpret_cast<uint16_t*>(out_v)[orow * cout + col] = (uint16_t)f2bf_rn(v);
            else out[orow * cout + col] = v;
          }
        }
      }
    }
  }
}

// dense_small_kernel: the same tile arithmetic for launches too small to hide latency (n < 8192 rows) — see its first comment.
template <int W_OUT_IN, int CIN, bool IN_BF16>
__device__ static inline void dense_small_body(const void* __restrict__ in_v, int64_t n,
                                               const float* __restrict__ W, int cout,
                                               const float* __restrict__ bias, const float* __restrict__ scale,
                                               const float* __restrict__ shift, int act,
                                               const void* __restrict__ residual_v, void* __restrict__ out_v, int io,
                                               const int32_t* __restrict__ n_dev, const int block_x, const int block_y) {
  // ONE memory round trip per wave: the live row count, the 16 input rows, the weights of all four 16-column tiles,
  // the epilogue vectors and the residual values are all requested before anything is waited for (the first version
  // walked a chain of five dependent round trips — count, rows, weights of pass 0, its epilogue vectors, weights of
  // pass 1 ... — and took 12-14 us on launches of a dozen workgroups, eight times per step: profiles/r02x_timeline.txt).
  // `n` is the capacity of the buffers, so rows between the live count and `n` may be read (never stored).
  const int64_t ncap = n;
  const int32_t* nd = n_dev ? n_dev : reinterpret_cast<const int32_t*>(W);      // every load below is unconditional:
  const int32_t nlive_raw = *nd;                                                // indices are clamped into the buffers and
  const int32_t nlive32 = n_dev ? nlive_raw : 0x7FFFFFFF;                       // null pointers replaced by W, so the code is
  const float* in = reinterpret_cast<const float*>(in_v);                       // one straight line of requests followed by
  float* out = reinterpret_cast<float*>(out_v);                                 // one wait (guarded loads compiled into a
  constexpr int KS = CIN / 16;                                                  // branch and a partial wait per load)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, g4 = lane >> 4;
  const int64_t row_base = ((int64_t)block_x * 4 + wave) * 16;
  if (row_base >= ncap) return;
  const int64_t row = min(row_base + l15, ncap - 1);
  float4 a[KS];
#pragma unroll
  for (int t = 0; t < KS; ++t) {
    if constexpr (IN_BF16) {
      const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(in_v) + row * CIN + 16 * t + 4 * g4);
      a[t] = make_float4(bf2f(h.x & 0xFFFFu), bf2f(h.x >> 16), bf2f(h.y & 0xFFFFu), bf2f(h.y >> 16));
    } else {
      a[t] = *reinterpret_cast<const float4*>(in + row * CIN + 16 * t + 4 * g4);
    }
  }
  const int ncol0 = block_y * 64;
  float4 b[4][KS];
  float bi[4], sc[4], sh[4];
  const float* bias_p = bias ? bias : W;
  const float* scale_p = scale ? scale : W;
  const float* shift_p = scale ? shift : W;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int col = min(ncol0 + nt * 16 + l15, cout - 1);
    bi[nt] = bias_p[col];
    sc[nt] = scale_p[col];
    sh[nt] = shift_p[col];
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      float4 v;
      if (W_OUT_IN) {
        v = *reinterpret_cast<const float4*>(W + (int64_t)col * CIN + 16 * t + 4 * g4);
      } else {
        const float* wp = W + (int64_t)(16 * t + 4 * g4) * cout + col;
        v.x = wp[0];
        v.y = wp[cout];
        v.z = wp[2 * (int64_t)cout];
        v.w = wp[3 * (int64_t)cout];
      }
      b[nt][t] = v;
    }
  }
  float res[4][4];
  {
    const bool has_res = residual_v != nullptr;
    const void* rp = has_res ? residual_v : static_cast<const void*>(W);
    int64_t ridx[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int col = min(ncol0 + nt * 16 + l15, cout - 1);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t orow = min(row_base + 4 * g4 + r, ncap - 1);
        ridx[nt][r] = has_res ? orow * cout + col : 0;
      }
    }
    if (io & 1) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) res[nt][r] = bf2f(reinterpret_cast<const uint16_t*>(rp)[ridx[nt][r]]);
    } else {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) res[nt][r] = reinterpret_cast<const float*>(rp)[ridx[nt][r]];
    }
  }
  if (!bias) bi[0] = bi[1] = bi[2] = bi[3] = 0.f;
  n = min((int64_t)nlive32, ncap);
  if (row_base >= n) return;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int col = ncol0 + nt * 16 + l15;
    if (ncol0 + nt * 16 >= cout) break;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].x, b[nt][t].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].y, b[nt][t].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].z, b[nt][t].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].w, b[nt][t].w, acc, 0, 0, 0);
    }
    if (col < cout) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t orow = row_base + 4 * g4 + r;
        if (orow < n) {
          float v = acc[r] + bi[nt];
          if (scale) v = v * sc[nt] + sh[nt];
          v = apply_act(v, act);
          if (residual_v) v += res[nt][r];
          if (io & 2) reinterpret_cast<uint16_t*>(out_v)[orow * cout + col] = (uint16_t)f2bf_rn(v);
          else out[orow * cout + col] = v;
        }
      }
    }
  }
}

// sample (scan) of row r: last b with boff[b] <= r
template <int W_OUT_IN, int CIN, bool IN_BF16>
__global__ __launch_bounds__(256) void dense_small_kernel(const void* __restrict__ in_v, int64_t n,
                                                    const float* __restrict__ W, int cout,
                                                    const float* __restrict__ bias, const float* __restrict__ scale,
                                                    const float* __restrict__ shift, int act,
                                                    const void* __restrict__ residual_v, void* __restrict__ out_v, int io,
                                                    const int32_t* __restrict__ n_dev) {
  dense_small_body<W_OUT_IN, CIN, IN_BF16>(in_v, n, W, cout, bias, scale, shift, act, residual_v, out_v, io, n_dev, (int)blockIdx.x,
                                           (int)blockIdx.y);
}
// Up to three INDEPENDENT small products of one shape in one launch (the three lateral 1x1 convolutions of MinkHead,
// models/minkgl.py:46-60: they depend on the trunk only, not on each other): block ranges of one grid, the body of the kernel above.
struct DenseGroup3 {
  const void* in[3];
  const float* W[3];
  void* out[3];
  const int32_t* n_dev[3];
  int64_t n[3];
  int bx0[4];                // first block of every problem along grid.x
};
template <int W_OUT_IN, int CIN, bool IN_BF16>
__global__ __launch_bounds__(256) void dense_small3_kernel(const DenseGroup3 g, int cout, int io) {
  const int b = (int)blockIdx.x;
  const int q = b >= g.bx0[2] ? 2 : (b >= g.bx0[1] ? 1 : 0);
  dense_small_body<W_OUT_IN, CIN, IN_BF16>(g.in[q], g.n[q], g.W[q], cout, nullptr, nullptr, nullptr, ACT_NONE, nullptr, g.out[q], io,
                                           g.n_dev[q], b - g.bx0[q], (int)blockIdx.y);
}

__device__ static inline int sample_of_row(const int32_t* __restrict__ boff, int B, int32_t r) {
  int lo = 0, hi = B;   // boff[lo] <= r < boff[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (boff[mid] <= r) lo = mid; else hi = mid;
  }
  return lo;
}

// The same product with the weights staged ONCE per workgroup into LDS as MFMA fragments
//   frag[(nt * CIN/16 + t) * 64 + lane] = W(col = 16 nt + (lane & 15), ci = 16 t + 4 (lane >> 4) .. +3)
// (whatever the caller's layout), workgroups persistent over the row tiles, operands swapped (D^T = W^T A^T) so that lane
// (row, g) owns four consecutive output columns: the epilogue is one 16-byte load / store per 16 columns.  The kernel
// above re-fetches its 32-64 KB of weights per 16-row tile (dword loads in the (in, out) layout) and stores single
// floats; this one runs the 1x1 convolutions of the heads 2x faster.  Same accumulation order => same results.
// Used when the fragments fit (Cin * Cout * 4 <= 96 KB) and Cout is a multiple of 16.
template <int W_OUT_IN, int CIN, bool IN_BF16>
__global__ __launch_bounds__(256) void dense_lds_kernel(const void* __restrict__ in_v, int64_t n,
                                                        const float* __restrict__ W, int cout,
                                                        const float* __restrict__ bias, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int act,
                                                        const void* __restrict__ residual_v, void* __restrict__ out_v, int io,
                                                        const int32_t* __restrict__ n_dev, int colblk,
                                                        const float* __restrict__ gate, const int32_t* __restrict__ boff, int B) {
  extern __shared__ __attribute__((aligned(16))) f32x4 dl_frags[];
  if (n_dev) n = min((int64_t)*n_dev, n);
  constexpr int KS = CIN / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g4 = lane >> 4;
  // grid.y splits the output columns into blocks of `colblk` (a multiple of 16) when all the fragments do not fit in LDS
  const int cb0 = blockIdx.y * colblk;
  const int NT = min(colblk, cout - cb0) >> 4, ncl = NT * 16;
  // eight fragments per thread in flight (one at a time made the staging a chain of up to 96 dependent round trips:
  // 24 us for the 96 KB block of the 192 -> 256 layer)
  auto frag_of = [&](int i) -> f32x4 {
    const int ln = i & 63, ft = i >> 6;
    const int nt = ft / KS, t = ft - nt * KS;
    const int col = cb0 + 16 * nt + (ln & 15), k0 = 16 * t + 4 * (ln >> 4);
    if (W_OUT_IN) return *reinterpret_cast<const f32x4*>(W + (int64_t)col * CIN + k0);
    const float* wp = W + (int64_t)k0 * cout + col;
    return (f32x4){wp[0], wp[cout], wp[2 * (int64_t)cout], wp[3 * (int64_t)cout]};
  };
  const int nfrag = NT * KS * 64;
  int i0 = tid;
  for (; i0 + 7 * 256 < nfrag; i0 += 8 * 256) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = frag_of(i0 + u * 256);
#pragma unroll
    for (int u = 0; u < 8; ++u) dl_frags[i0 + u * 256] = v[u];
  }
  for (; i0 < nfrag; i0 += 256) dl_frags[i0] = frag_of(i0);
  __syncthreads();
  // bias / folded-BN vectors of the epilogue -> LDS (behind the fragments): they were three dependent L1 round trips per
  // 16 output columns of every tile
  float* const s_vec = reinterpret_cast<float*>(dl_frags + (int64_t)NT * KS * 64);      // [3][cout]: bias, scale, shift
  for (int i = tid; i < ncl; i += 256) {
    s_vec[i] = bias ? bias[cb0 + i] : 0.f;
    s_vec[ncl + i] = scale ? scale[cb0 + i] : 1.f;
    s_vec[2 * ncl + i] = scale ? shift[cb0 + i] : 0.f;
  }
  __syncthreads();
  const int64_t ntiles = (n + 15) >> 4;
  const int64_t tstep = (int64_t)gridDim.x * 4;
  auto load_rows = [&](int64_t tile, f32x4 (&a)[KS]) {      // the 16 input rows of a tile (zeros past the end)
    const int64_t row = tile * 16 + l15;
#pragma unroll
    for (int t = 0; t < KS; ++t) {
      a[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (tile < ntiles && row < n) {
        if constexpr (IN_BF16) {
          const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(in_v) + row * CIN + 16 * t + 4 * g4);
          a[t] = (f32x4){bf2f(h.x & 0xFFFFu), bf2f(h.x >> 16), bf2f(h.y & 0xFFFFu), bf2f(h.y >> 16)};
        } else {
          a[t] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(in_v) + row * CIN + 16 * t + 4 * g4);
        }
      }
    }
  };
  f32x4 a[KS], an[KS];
  int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  load_rows(tile, a);
  for (; tile < ntiles; tile += tstep) {
    const int64_t row = tile * 16 + l15;
    const bool ok = row < n;
    load_rows(tile + tstep, an);                             // next tile's rows in flight while this one is computed
    // gate != null: the ECA tail of a block with a 1x1 downsample branch (layers/eca_block.py:66-73) fused into this launch —
    // out = relu(residual * gate[sample of the row] + this layer's output): the downsample output never goes to memory
    const int64_t gbase = (gate && ok) ? (int64_t)sample_of_row(boff, B, (int32_t)row) * cout : 0;
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const f32x4* fr = dl_frags + (int64_t)nt * KS * 64 + lane;
#pragma unroll
      for (int t = 0; t < KS; ++t) {
        const f32x4 wf = fr[t * 64];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u], a[t][u], acc, 0, 0, 0);
      }
      if (ok) {
        const int cl = 16 * nt + 4 * g4, c0 = cb0 + cl;
        f32x4 v = acc + *reinterpret_cast<const f32x4*>(s_vec + cl);
        v = v * *reinterpret_cast<const f32x4*>(s_vec + ncl + cl) + *reinterpret_cast<const f32x4*>(s_vec + 2 * ncl + cl);
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = apply_act(v[u], act);
        if (residual_v) {
          f32x4 r;
          if (io & 1) {
            const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(residual_v) + row * cout + c0);
            r = (f32x4){bf2f(h.x & 0xFFFFu), bf2f(h.x >> 16), bf2f(h.y & 0xFFFFu), bf2f(h.y >> 16)};
          } else {
            r = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(residual_v) + row * cout + c0);
          }
          if (gate) {
            const f32x4 gq = *reinterpret_cast<const f32x4*>(gate + gbase + c0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              // (bf16 maps: the downsample output was a bf16 map between the two launches this replaces; same rounding here)
              const float d = (io & 2) ? bf2f(f2bf_rn(v[u])) : v[u];
              v[u] = fmaxf(r[u] * gq[u] + d, 0.f);           // the expression of eca_apply_kernel
            }
          } else {
            v += r;
          }
        }
        if (io & 2) {
          *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out_v) + row * cout + c0) =
              make_uint2(f2bf_rn(v[0]) | (f2bf_rn(v[1]) << 16), f2bf_rn(v[2]) | (f2bf_rn(v[3]) << 16));
        } else {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out_v) + row * cout + c0) = v;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < KS; ++t) a[t] = an[t];
  }
}

__global__ void dense_any_kernel(const float* __restrict__ in, int64_t total, int cin, const float* __restrict__ W,
                                 int w_out_in, int cout, const float* __restrict__ bias, int act, float* __restrict__ out,
                                 const int32_t* __restrict__ n_dev) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_dev) total = min((int64_t)*n_dev * cout, total);
  if (i >= total) return;
  const int64_t r = i / cout;
  const int co = (int)(i - r * cout);
  float s = bias ? bias[co] : 0.f;
  for (int ci = 0; ci < cin; ++ci) s = fmaf(in[r * cin + ci], w_out_in ? W[(int64_t)co * cin + ci] : W[(int64_t)ci * cout + co], s);
  out[i] = apply_act(s, act);
}

// true: dense_forward_ex(..., gate, boff, B) can fuse the gated-residual ECA tail for this shape (the LDS-staged kernel runs it)
bool dense_gate_fusable(int64_t n, int cin, int cout) {
  const bool plan = cin == 32 || cin == 64 || cin == 96 || cin == 128 || cin == 192 || cin == 256;
  return plan && cout % 16 == 0 && n >= 8192 && (size_t)cin * cout * sizeof(float) <= 96 * 1024;
}

int dense_forward_ex(const void* in, int in_bf16, int64_t n, int cin, const float* W, int w_out_in, int cout,
                     const float* bias, const float* scale, const float* shift, int act, const void* residual, int res_bf16,
                     void* out, int out_bf16, hipStream_t stream, const int32_t* n_dev, const float* gate, const int32_t* boff,
                     int B) {
  if (n == 0) return EGONN_OK;
  EGONN_REQUIRE(!gate || (residual && boff && B >= 1 && dense_gate_fusable(n, cin, cout)), EGONN_ERR_INVALID,
                "dense: the fused gated residual needs the LDS-staged kernel (%lld rows, %d->%d)", (long long)n, cin, cout);
  const dim3 grid((unsigned)cdiv(n, 64), (unsigned)cdiv(cout, 64));
  const int io = (res_bf16 ? 1 : 0) | (out_bf16 ? 2 : 0);
  const size_t frag_bytes = (size_t)cin * cout * sizeof(float);
  // weights resident in LDS, workgroups persistent over the row tiles; weight sets above 96 KB (the 192 -> 256 layer of the
  // global decoder) are split over grid.y into column blocks that fit
  const int ysplit = (int)cdiv((int64_t)frag_bytes, 96 * 1024);
  const int colblk = (int)(cdiv(cdiv(cout, ysplit), 16) * 16);
  // measured (batch 16 / 64): from 8192 rows on the staged weights win over dense_small_kernel; the split blocks replace
  // the 64-column kernel that re-reads its 48 KB weight slice per wave (17 / 56 us -> 18 / 41 us for 192 -> 256)
  if (cout % 16 == 0 && n >= (ysplit == 1 ? 8192 : 2048)) {
    const size_t blk_bytes = (size_t)cin * colblk * sizeof(float);
    const int ny = (int)cdiv(cout, colblk);
    const unsigned g1 = (unsigned)std::min<int64_t>(cdiv(cdiv(n, 16), 4), (blk_bytes > 72 * 1024 ? 256 : 512) / ny);
#define EGONN_DENSE_LDS_LAUNCH(WOI, CI, INB)                                                                           \
  {                                                                                                                   \
    static AttrOnce attr_done;                                                                                    \
    if (attr_done.need()) {                                                                                                 \
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_lds_kernel<WOI, CI, INB>),                   \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                         \
      attr_done.mark();                                                                                                \
    }                                                                                                                 \
    hipLaunchKernelGGL((dense_lds_kernel<WOI, CI, INB>), dim3(g1, ny), dim3(256), blk_bytes + 3 * (size_t)colblk * sizeof(float), stream, in, n, W, cout, bias, scale, \
                       shift, act, residual, out, io, n_dev, colblk, gate, boff, B);                                  \
  }
#define EGONN_DENSE_LDS_CASE(CI)                                                                                      \
  if (cin == CI) {                                                                                                    \
    if (w_out_in) { if (in_bf16) EGONN_DENSE_LDS_LAUNCH(1, CI, true) else EGONN_DENSE_LDS_LAUNCH(1, CI, false) }       \
    else          { if (in_bf16) EGONN_DENSE_LDS_LAUNCH(0, CI, true) else EGONN_DENSE_LDS_LAUNCH(0, CI, false) }       \
    HIP_CHECK(hipGetLastError());                                                                                     \
    return EGONN_OK;                                                                                                  \
  }
    EGONN_DENSE_LDS_CASE(32)
    EGONN_DENSE_LDS_CASE(64)
    EGONN_DENSE_LDS_CASE(96)
    EGONN_DENSE_LDS_CASE(128)
    EGONN_DENSE_LDS_CASE(192)
    EGONN_DENSE_LDS_CASE(256)
#undef EGONN_DENSE_LDS_CASE
#undef EGONN_DENSE_LDS_LAUNCH
  }
#define EGONN_DENSE_LAUNCH(WOI, CI, INB)                                                                              \
  if (n < 8192 && CI <= 128)                                                                                          \
    hipLaunchKernelGGL((dense_small_kernel<WOI, CI, INB>), grid, dim3(256), 0, stream, in, n, W, cout, bias, scale, shift, act, \
                       residual, out, io, n_dev);                                                                     \
  else                                                                                                                \
    hipLaunchKernelGGL((dense_kernel<WOI, CI, INB>), grid, dim3(256), 0, stream, in, n, W, cout, bias, scale, shift, act, \
                       residual, out, io, n_dev)
#define EGONN_DENSE_CASE(CI)                                                                                       \
  if (cin == CI) {                                                                                                 \
    if (w_out_in) { if (in_bf16) EGONN_DENSE_LAUNCH(1, CI, true); else EGONN_DENSE_LAUNCH(1, CI, false); }         \
    else          { if (in_bf16) EGONN_DENSE_LAUNCH(0, CI, true); else EGONN_DENSE_LAUNCH(0, CI, false); }         \
    HIP_CHECK(hipGetLastError());                                                                                  \
    return EGONN_OK;                                                                                               \
  }
  EGONN_DENSE_CASE(32)
  EGONN_DENSE_CASE(64)
  EGONN_DENSE_CASE(96)
  EGONN_DENSE_CASE(128)
  EGONN_DENSE_CASE(192)
  EGONN_DENSE_CASE(256)
#undef EGONN_DENSE_CASE
#undef EGONN_DENSE_LAUNCH
  // any other width (the 3- and 1-channel gradients of the keypoint / sigma regressors): plain kernel, thread per output
  EGONN_REQUIRE(!scale && !shift && !residual && !in_bf16 && !out_bf16, EGONN_ERR_INVALID, "dense: cin=%d has no fused epilogue", cin);
  const int64_t total = n * cout;
  hipLaunchKernelGGL(dense_any_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream,
                     reinterpret_cast<const float*>(in), total, cin, W, w_out_in, cout, bias, act, reinterpret_cast<float*>(out),
                     n_dev);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}
// Three (n_i, 128) @ (128, 128) products ((cin, cout) kernels, no bias / BN / activation) in ONE launch of dense_small3_kernel;
// fp32 rows in and out.  Every n_i must be below the 8192 rows up to which the small kernel is the product choice.
int dense_small_group3(const float* const* in, const int64_t* n, const int32_t* const* n_dev, const float* const* W, float* const* out,
                       hipStream_t stream) {
  DenseGroup3 g;
  int bx = 0;
  for (int q = 0; q < 3; ++q) {
    EGONN_REQUIRE(n[q] < 8192, EGONN_ERR_INVALID, "dense(group): %lld rows", (long long)n[q]);
    g.in[q] = in[q]; g.W[q] = W[q]; g.out[q] = out[q]; g.n_dev[q] = n_dev[q]; g.n[q] = n[q];
    g.bx0[q] = bx;
    bx += (int)cdiv(std::max<int64_t>(n[q], 1), 64);
  }
  g.bx0[3] = bx;
  hipLaunchKernelGGL((dense_small3_kernel<0, 128, false>), dim3((unsigned)bx, 2), dim3(256), 0, stream, g, 128, 0);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

int dense_forward(const float* in, int64_t n, int cin, const float* W, int w_out_in, int cout, const float* bias,
                  const float* scale, const float* shift, int act, const float* residual, float* out,
                  hipStream_t stream) {
  return dense_forward_ex(in, 0, n, cin, W, w_out_in, cout, bias, scale, shift, act, residual, 0, out, 0, stream, nullptr);
}

// ------------------------------------------------------------------ BatchNorm folding (eval mode)
__global__ void bn_fold_kernel(const float* __restrict__ w, const float* __restrict__ b,
                               const float* __restrict__ rm, const float* __restrict__ rv, float eps, int c,
                               float* __restrict__ scale, float* __restrict__ shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  const float s = w[i] / sqrtf(rv[i] + eps);
  scale[i] = s;
  shift[i] = b[i] - rm[i] * s;
}
int bn_fold(const float* w, const float* b, const float* rm, const float* rv, float eps, int c, float* scale,
            float* shift, hipStream_t stream) {
  hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)cdiv(c, 128)), dim3(128), 0, stream, w, b, rm, rv, eps, c, scale,
                     shift);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------ row gather (features -> sorted row order)
__global__ void gather_rows_kernel(const float* __restrict__ in, const int32_t* __restrict__ perm, int64_t n, int c,
                                   float* __restrict__ out, const int32_t* __restrict__ n_dev) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_dev) n = min((int64_t)*n_dev, n);
  if (t >= n * c) return;
  const int64_t r = t / c;
  const int ch = (int)(t - r * c);
  out[t] = in[(int64_t)perm[r] * c + ch];
}
int gather_rows(const float* in, const int32_t* perm, int64_t n, int c, float* out, hipStream_t stream, const int32_t* n_dev) {
  if (n == 0) return EGONN_OK;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)cdiv(n * c, 256)), dim3(256), 0, stream, in, perm, n, c, out, n_dev);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------ per-sample column sums (deterministic)
// grid (SEG_CHUNKS, B): block (ch, b) sums rows [s + ch*len/CH, s + (ch+1)*len/CH) of sample b.
__global__ __launch_bounds__(256) void segment_partial_kernel(const float* __restrict__ in,
                                                               const int32_t* __restrict__ boff, int c, int pow_mode,
                                                               const float* __restrict__ pexp,
                                                               float* __restrict__ partial) {
  __shared__ float red[256];
  const int b = blockIdx.y, ch = blockIdx.x;
  const int32_t s = boff[b], e = boff[b + 1];
  const int64_t len = e - s;
  const int32_t r0 = s + (int32_t)(len * ch / SEG_CHUNKS), r1 = s + (int32_t)(len * (ch + 1) / SEG_CHUNKS);
  const int tid = threadIdx.x;
  const float p = pow_mode == 1 ? pexp[0] : 1.f;           // pow_mode: 0 = sum, 1 = sum of clamp(x)^p (GeM), 2 = max (MAC)
  // c <= 256: thread -> (row lane, channel); for c < 256 several rows advance in parallel
  const int rl = tid / c, cidx = tid - rl * c, nrl = 256 / c;
  float acc = pow_mode == 2 ? -INFINITY : 0.f;
  if (rl < nrl) {
    for (int32_t r = r0 + rl; r < r1; r += nrl) {
      float v = in[(int64_t)r * c + cidx];
      if (pow_mode == 1) v = powf(fmaxf(v, 1e-6f), p);
      acc = pow_mode == 2 ? fmaxf(acc, v) : acc + v;
    }
  }
  red[tid] = acc;
  __syncthreads();
  if (tid < c) {
    float sum = pow_mode == 2 ? -INFINITY : 0.f;
    for (int k = 0; k < nrl; ++k) sum = pow_mode == 2 ? fmaxf(sum, red[k * c + tid]) : sum + red[k * c + tid];
    partial[((int64_t)b * SEG_CHUNKS + ch) * c + tid] = sum;
  }
}
int segment_partial_sums(const float* in, const int32_t* boff, int B, int c, int pow_mode, const float* p,
                         float* partial, hipStream_t stream) {
  EGONN_REQUIRE(c >= 1 && c <= 256 && 256 % c == 0, EGONN_ERR_INVALID, "segment sums: %d channels unsupported", c);
  hipLaunchKernelGGL(segment_partial_kernel, dim3(SEG_CHUNKS, B), dim3(256), 0, stream, in, boff, c, pow_mode, p,
                     partial);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------ ECA: gate = sigmoid(conv1d_k(mean)) ; out = relu(x*gate + res)
__global__ void eca_gate_kernel(const float* __restrict__ partial, const int32_t* __restrict__ boff, int c,
                                const float* __restrict__ wconv, int ksize, float* __restrict__ gate) {
  extern __shared__ float mean[];
  const int b = blockIdx.x, t = threadIdx.x;
  const int32_t cntr = boff[b + 1] - boff[b];
  if (t < c) {
    float s = 0.f;
    for (int ch = 0; ch < SEG_CHUNKS; ++ch) s += partial[((int64_t)b * SEG_CHUNKS + ch) * c + t];
    mean[t] = cntr > 0 ? s / (float)cntr : 0.f;
  }
  __syncthreads();
  if (t < c) {
    const int pad = (ksize - 1) / 2;
    float y = 0.f;
    for (int j = 0; j < ksize; ++j) {
      const int q = t + j - pad;
      if (q >= 0 && q < c) y += wconv[j] * mean[q];
    }
    gate[(int64_t)b * c + t] = 1.f / (1.f + expf(-y));
  }
}


// thread = 16 bytes of a row (4 fp32 or 8 bf16 channels): full-width accesses for both precisions (8-byte accesses run at
// 0.5-0.7 of the 16-byte rate); cq = channels / (16 bytes' worth) is a power of two
template <bool BF16>
__global__ void eca_apply_kernel(const void* __restrict__ x, const void* __restrict__ res,
                                 const float* __restrict__ gate, const int32_t* __restrict__ boff, int B, int64_t n,
                                 int cq_shift, void* __restrict__ out) {
  constexpr int CPT = BF16 ? 8 : 4;                      // channels per thread
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  n = min((int64_t)boff[B], n);                          // rows in use (reserved plans size the grid for the capacity)
  if (t >= (n << cq_shift)) return;
  const int32_t r = (int32_t)(t >> cq_shift);
  const int q = (int)(t - ((int64_t)r << cq_shift));
  const int b = sample_of_row(boff, B, r);
  const float* gp = gate + (((int64_t)b << cq_shift) + q) * CPT;
  float xv[CPT], rv[CPT], g[CPT];
  if constexpr (BF16) {
    const uint4 xh = reinterpret_cast<const uint4*>(x)[t], rh = reinterpret_cast<const uint4*>(res)[t];
    const uint32_t xw[4] = {xh.x, xh.y, xh.z, xh.w}, rw[4] = {rh.x, rh.y, rh.z, rh.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xv[2 * i] = bf2f(xw[i] & 0xFFFFu); xv[2 * i + 1] = bf2f(xw[i] >> 16);
      rv[2 * i] = bf2f(rw[i] & 0xFFFFu); rv[2 * i + 1] = bf2f(rw[i] >> 16);
    }
    const float4 g0 = reinterpret_cast<const float4*>(gp)[0], g1 = reinterpret_cast<const float4*>(gp)[1];
    g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
  } else {
    const float4 xf = reinterpret_cast<const float4*>(x)[t], rf = reinterpret_cast<const float4*>(res)[t];
    const float4 g0 = reinterpret_cast<const float4*>(gp)[0];
    xv[0] = xf.x; xv[1] = xf.y; xv[2] = xf.z; xv[3] = xf.w;
    rv[0] = rf.x; rv[1] = rf.y; rv[2] = rf.z; rv[3] = rf.w;
    g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w;
  }
  float o[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) o[i] = fmaxf(xv[i] * g[i] + rv[i], 0.f);
  if constexpr (BF16) {
    uint4 oh;
    oh.x = f2bf_rn(o[0]) | (f2bf_rn(o[1]) << 16);
    oh.y = f2bf_rn(o[2]) | (f2bf_rn(o[3]) << 16);
    oh.z = f2bf_rn(o[4]) | (f2bf_rn(o[5]) << 16);
    oh.w = f2bf_rn(o[6]) | (f2bf_rn(o[7]) << 16);
    reinterpret_cast<uint4*>(out)[t] = oh;
  } else {
    reinterpret_cast<float4*>(out)[t] = make_float4(o[0], o[1], o[2], o[3]);
  }
}
__global__ void bf16_to_f32_kernel(const uint16_t* __restrict__ in, int64_t n, float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) out[t] = bf2f(in[t]);
}
int convert_bf16_to_f32(const void* in, int64_t n, float* out, hipStream_t stream) {
  if (n == 0) return EGONN_OK;
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream,
                     reinterpret_cast<const uint16_t*>(in), n, out);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

int eca_apply(const float* x, const float* res, const float* partial, const int32_t* boff, int B, int64_t n, int c,
              const float* wconv, int ksize, float* out, hipStream_t stream) {
  EGONN_REQUIRE(c >= 4 && c <= 256 && c % 4 == 0, EGONN_ERR_INVALID, "eca: %d channels unsupported (4..256, multiple of 4)", c);
  if (n == 0) return EGONN_OK;
  // gate lives right behind the partial sums: partial[B*SEG_CHUNKS*c] | gate[B*c]
  float* gate = const_cast<float*>(partial) + (int64_t)B * SEG_CHUNKS * c;
  hipLaunchKernelGGL(eca_gate_kernel, dim3(B), dim3(256), c * sizeof(float), stream, partial, boff, c, wconv, ksize,
                     gate);
  const int c4 = c / 4;
  EGONN_REQUIRE((c4 & (c4 - 1)) == 0, EGONN_ERR_INVALID, "eca_apply: %d channels (a power of two expected)", c);
  hipLaunchKernelGGL(eca_apply_kernel<false>, dim3((unsigned)cdiv(n * c4, 256)), dim3(256), 0, stream, x, res, gate, boff, B,
                     n, __builtin_ctz(c4), out);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ECA gate straight from the per-group column sums of the conv2 epilogue (sconv.hip): the groups of sample b are
// rg.meta[1+b] .. rg.meta[2+b] (a window never straddles two samples), summed in fixed order => deterministic.
__global__ __launch_bounds__(1024) void eca_gate_groups_kernel(const float* __restrict__ psum,
                                                                const uint32_t* __restrict__ gmask,
                                                                const int32_t* __restrict__ meta,
                                                                const int32_t* __restrict__ boff, int c,
                                                                const float* __restrict__ wconv, int ksize,
                                                                float* __restrict__ gate) {
  __shared__ float red[1024];
  __shared__ float mean[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int g0 = meta[1 + b], g1 = meta[2 + b];
  const int32_t cntr = boff[b + 1] - boff[b];
  const int nsl = 1024 / c, sl = tid / c, ch = tid - sl * c;    // c <= 256, power of two: 4..32 slices of groups
  // fixed order per slice; the loads of 8 groups are in flight together, and the sums are read whether or not the group
  // is flagged (the buffer covers every group; an unflagged group's value is dropped by the select): with the load behind
  // the flag every group cost two dependent round trips, 16 us for the 800 groups per scan of level 1.
  float acc = 0.f;
  {
    constexpr int U = 8;
    int g = g0 + sl;
    for (; g + (U - 1) * nsl < g1; g += U * nsl) {
      float v[U];
      uint32_t m[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        m[u] = gmask[g + u * nsl];
        v[u] = psum[(int64_t)(g + u * nsl) * c + ch];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc += (m[u] >> 31) ? v[u] : 0.f;
    }
    for (; g < g1; g += nsl) {
      const uint32_t m = gmask[g];
      const float v = psum[(int64_t)g * c + ch];
      acc += (m >> 31) ? v : 0.f;
    }
  }
  red[tid] = acc;
  __syncthreads();
  if (tid < c) {
    float s = 0.f;
    for (int k = 0; k < nsl; ++k) s += red[k * c + tid];
    mean[tid] = cntr > 0 ? s / (float)cntr : 0.f;
  }
  __syncthreads();
  if (tid < c) {
    const int pad = (ksize - 1) / 2;
    float y = 0.f;
    for (int j = 0; j < ksize; ++j) {
      const int q = tid + j - pad;
      if (q >= 0 && q < c) y += wconv[j] * mean[q];
    }
    gate[(int64_t)b * c + tid] = 1.f / (1.f + expf(-y));
  }
}
int eca_gate_groups(const float* psum, const RowGroups& rg, const int32_t* boff, int B, int c, const float* wconv, int ksize,
                    float* gate, hipStream_t stream) {
  EGONN_REQUIRE(c >= 32 && c <= 256 && 256 % c == 0, EGONN_ERR_INVALID, "eca gate: %d channels unsupported", c);
  hipLaunchKernelGGL(eca_gate_groups_kernel, dim3(B), dim3(1024), 0, stream, psum, rg.gmask, rg.meta, boff, c, wconv, ksize,
                     gate);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}
int eca_apply_gate(const void* x, const void* res, const float* gate, const int32_t* boff, int B, int64_t n, int c,
                   void* out, int bf16, hipStream_t stream) {
  if (n == 0) return EGONN_OK;
  const int cq = bf16 ? c / 8 : c / 4;                   // 16-byte pieces per row
  EGONN_REQUIRE(cq >= 1 && (cq & (cq - 1)) == 0, EGONN_ERR_INVALID, "eca_apply: %d channels (a power of two expected)", c);
  if (bf16)
    hipLaunchKernelGGL(eca_apply_kernel<true>, dim3((unsigned)cdiv(n * cq, 256)), dim3(256), 0, stream, x, res, gate, boff, B, n,
                       __builtin_ctz(cq), out);
  else
    hipLaunchKernelGGL(eca_apply_kernel<false>, dim3((unsigned)cdiv(n * cq, 256)), dim3(256), 0, stream, x, res, gate, boff, B, n,
                       __builtin_ctz(cq), out);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

__global__ void add_act_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, int relu,
                               float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const float v = a[t] + b[t];
  out[t] = relu ? fmaxf(v, 0.f) : v;
}
int add_act(const float* a, const float* b, int64_t n, int relu, float* out, hipStream_t stream) {
  if (n == 0) return EGONN_OK;
  hipLaunchKernelGGL(add_act_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, a, b, n, relu, out);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------ GeM finish
// pexp != NULL: GeM (mean^(1/p)); pexp == NULL: mode 0 = SPoC (mean, MinkowskiGlobalAvgPooling), 2 = MAC (max)
__global__ void gem_finish_kernel(const float* __restrict__ partial, const int32_t* __restrict__ boff, int c,
                                  const float* __restrict__ pexp, int mode, float* __restrict__ out) {
  const int b = blockIdx.x, t = threadIdx.x;
  if (t >= c) return;
  const int32_t cntr = boff[b + 1] - boff[b];
  if (mode == 2) {
    float s = -INFINITY;
    for (int ch = 0; ch < SEG_CHUNKS; ++ch) s = fmaxf(s, partial[((int64_t)b * SEG_CHUNKS + ch) * c + t]);
    out[(int64_t)b * c + t] = cntr > 0 ? s : 0.f;
    return;
  }
  float s = 0.f;
  for (int ch = 0; ch < SEG_CHUNKS; ++ch) s += partial[((int64_t)b * SEG_CHUNKS + ch) * c + t];
  const float m = cntr > 0 ? s / (float)cntr : 0.f;
  out[(int64_t)b * c + t] = pexp ? powf(m, 1.f / pexp[0]) : m;
}
int gem_finish(const float* partial, const int32_t* boff, int B, int c, const float* p, float* out,
               hipStream_t stream) {
  EGONN_REQUIRE(c >= 1 && c <= 256, EGONN_ERR_INVALID, "gem: %d channels unsupported (1..256)", c);
  hipLaunchKernelGGL(gem_finish_kernel, dim3(B), dim3(256), 0, stream, partial, boff, c, p, 0, out);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

int pool_finish(const float* partial, const int32_t* boff, int B, int c, int mode, float* out, hipStream_t stream) {
  EGONN_REQUIRE(c >= 1 && c <= 256 && (mode == 0 || mode == 2), EGONN_ERR_INVALID, "pooling: bad arguments");
  hipLaunchKernelGGL(gem_finish_kernel, dim3(B), dim3(256), 0, stream, partial, boff, c, (const float*)nullptr, mode, out);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------ row L2 normalisation (F.normalize eps=1e-12)
__global__ void l2norm_kernel(float* __restrict__ x, int64_t n, int c, const int32_t* __restrict__ n_dev) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (n_dev) n = min((int64_t)*n_dev, n);
  if (row >= n) return;
  float* p = x + row * c;
  float ss = 0.f;
  for (int i = lane; i < c; i += 64) ss += p[i] * p[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
  for (int i = lane; i < c; i += 64) p[i] *= inv;
}
int l2_normalize_rows(float* x, int64_t n, int c, hipStream_t stream, const int32_t* n_dev) {
  if (n == 0) return EGONN_OK;
  hipLaunchKernelGGL(l2norm_kernel, dim3((unsigned)cdiv(n, 4)), dim3(256), 0, stream, x, n, c, n_dev);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------ keypoint positions
// Quantizer.keypoint_position (datasets/quantization.py:60-72 polar, 93-103 Cartesian) of one super-voxel
__device__ static inline void keypoint_position(uint64_t key, int level, int cb, float ox, float oy, float oz, int mode,
                                                float s0, float s1, float s2, float* __restrict__ out3) {
  const int cbL = cb - level;
  const uint64_t mort = key & ((1ull << (3 * cbL)) - 1);
  const int32_t bias = 1 << (cb - 1);
  const float cx = (float)(((int32_t)compact1by2(mort) << level) - bias);
  const float cy = (float)(((int32_t)compact1by2(mort >> 1) << level) - bias);
  const float cz = (float)(((int32_t)compact1by2(mort >> 2) << level) - bias);
  const float stride = (float)(1 << level);
  // (C + 0.5) * q + offset * (stride * q) / 2, evaluated like the reference (no contraction)
  const float px = __fadd_rn(__fmul_rn(__fadd_rn(cx, 0.5f), s0), __fdiv_rn(__fmul_rn(ox, __fmul_rn(stride, s0)), 2.f));
  const float py = __fadd_rn(__fmul_rn(__fadd_rn(cy, 0.5f), s1), __fdiv_rn(__fmul_rn(oy, __fmul_rn(stride, s1)), 2.f));
  const float pz = __fadd_rn(__fmul_rn(__fadd_rn(cz, 0.5f), s2), __fdiv_rn(__fmul_rn(oz, __fmul_rn(stride, s2)), 2.f));
  if (mode == 0) {
    out3[0] = px; out3[1] = py; out3[2] = pz;
  } else {
    // polar -> cartesian (reference quantization.py:46-53): theta = pi * (deg - 180) / 180
    const float theta = __fdiv_rn(__fmul_rn(3.14159265358979323846f, __fadd_rn(px, -180.f)), 180.f);
    out3[0] = cosf(theta) * py;
    out3[1] = sinf(theta) * py;
    out3[2] = pz;
  }
}

__global__ void keypoint_kernel(const uint64_t* __restrict__ keys, int64_t n, int level, int cb,
                                const float* __restrict__ offs, int mode, float s0, float s1, float s2, int ignore,
                                float* __restrict__ out, const int32_t* __restrict__ n_dev) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_dev) n = min((int64_t)*n_dev, n);
  if (i >= n) return;
  const float kox = ignore ? 0.f : offs[3 * i + 0], koy = ignore ? 0.f : offs[3 * i + 1],
              koz = ignore ? 0.f : offs[3 * i + 2];
  keypoint_position(keys[i], level, cb, kox, koy, koz, mode, s0, s1, s2, out + 3 * i);
}
#if 0
  const int cbL = cb - level;
  const uint64_t mort = keys[i] & ((1ull << (3 * cbL)) - 1);
  const int32_t bias = 1 << (cb - 1);
  const float cx = (float)(((int32_t)compact1by2(mort) << level) - bias);
  const float cy = (float)(((int32_t)compact1by2(mort >> 1) << level) - bias);
  const float cz = (float)(((int32_t)compact1by2(mort >> 2) << level) - bias);
  const float stride = (float)(1 << level);
  const float ox = ignore ? 0.f : offs[3 * i + 0], oy = ignore ? 0.f : offs[3 * i + 1],
              oz = ignore ? 0.f : offs[3 * i + 2];
  // (C + 0.5) * q + offset * (stride * q) / 2, evaluated like the reference (no contraction)
  const float px = __fadd_rn(__fmul_rn(__fadd_rn(cx, 0.5f), s0), __fdiv_rn(__fmul_rn(ox, __fmul_rn(stride, s0)), 2.f));
  const float py = __fadd_rn(__fmul_rn(__fadd_rn(cy, 0.5f), s1), __fdiv_rn(__fmul_rn(oy, __fmul_rn(stride, s1)), 2.f));
  const float pz = __fadd_rn(__fmul_rn(__fadd_rn(cz, 0.5f), s2), __fdiv_rn(__fmul_rn(oz, __fmul_rn(stride, s2)), 2.f));
  if (mode == 0) {
    out[3 * i + 0] = px; out[3 * i + 1] = py; out[3 * i + 2] = pz;
  } else {
    // polar -> cartesian (reference quantization.py:46-53): theta = pi * (deg - 180) / 180
    const float theta = __fdiv_rn(__fmul_rn(3.14159265358979323846f, __fadd_rn(px, -180.f)), 180.f);
    out[3 * i + 0] = cosf(theta) * py;
    out[3 * i + 1] = sinf(theta) * py;
    out[3 * i + 2] = pz;
  }
}
#endif
int keypoint_positions(const uint64_t* keys, int64_t n, int level, int cb, const float* offsets, int mode,
                       const float* step, int ignore_offsets, float* out, hipStream_t stream, const int32_t* n_dev) {
  if (n == 0) return EGONN_OK;
  const float s0 = step[0], s1 = mode ? step[1] : step[0], s2 = mode ? step[2] : step[0];
  hipLaunchKernelGGL(keypoint_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, keys, n, level, cb, offsets,
                     mode, s0, s1, s2, ignore_offsets, out, n_dev);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------ the three local heads in one launch
// DescriptorDecoder (Linear 64->96, ReLU, Linear 96->128, L2 normalise), KeypointRegressor (64->32, ReLU, 32->3, tanh,
// Quantizer.keypoint_position) and SigmaRegressor (64->32, ReLU, 32->1, softplus) of models/minkgl.py:175-225,287-308
// read the local feature map ONCE.  A wave owns 16 rows; operands are fed swapped (D^T = W^T X^T), so a lane ends up with
// four consecutive hidden units of its row — exactly the B fragment of the next layer's MFMA (contraction index
// 16t + 4g + u), i.e. the MLPs chain in registers.  Weights are read in nn.Linear (out,in) layout, one float4 per lane.
struct LocalHeadsArgs {
  const float* x;                                          // (n, 64)
  const float *lw, *lres;                                  // non-null: the head's lateral 1x1 convolution runs here first —
                                                           // x := x @ lw + lres (lw (64,64) in (cin, cout) layout, lres (n,64))
  const float *dw0, *db0, *dw1, *db1;                      // descriptor decoder: (96,64),(96),(128,96),(128)
  const float *kw0, *kb0, *kw1, *kb1;                      // keypoint regressor: (32,64),(32),(3,32),(3)
  const float *sw0, *sb0, *sw1, *sb1;                      // sigma regressor:    (32,64),(32),(1,32),(1)
  const uint64_t* keys;                                    // level-3 keys (super-voxel coordinates)
  const int32_t* n_dev;                                    // device row count (nullable)
  float *out_desc, *out_kp, *out_sigma;
  int64_t n;
  int level, cb, mode, ignore_offsets;
  int in_bf16;                                             // x and lres are bf16 maps (bf16 feature maps; only with lw)
  float s0, s1, s2;
};
// The six weight matrices (92 KB as MFMA fragments) are staged ONCE per workgroup into LDS in fragment order
//   frag[(nt * CIN/16 + t) * 64 + lane] = W[16 nt + (lane & 15)][16 t + 4 (lane >> 4) .. +3]      (rows >= rows_w: zeros)
// so that a wave's B operands are lane-linear ds_read_b128 (conflict-free) instead of 92 KB of dependent global loads per
// 16-row tile — the first version spent 60 us on 1.4 GFLOP because every wave re-fetched every weight through L1.
template <int CIN, int NT>
__device__ static inline void stage_frags(const float* __restrict__ W, int rows_w, f32x4* __restrict__ dst, int tid, int nthreads) {
  constexpr int T = CIN / 16;
  for (int i = tid; i < NT * T * 64; i += nthreads) {
    const int lane = i & 63, ft = i >> 6;
    const int nt = ft / T, t = ft - nt * T;
    const int unit = 16 * nt + (lane & 15);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (unit < rows_w) v = *reinterpret_cast<const f32x4*>(W + (int64_t)unit * CIN + 16 * t + 4 * (lane >> 4));
    dst[i] = v;
  }
}
template <int CIN, int NT>   // out[nt] = sum_k W[16nt + l15][k] * in[k]  for the 16 rows of the wave (swapped operands)
__device__ static inline void mlp_layer(const f32x4* __restrict__ frags, const f32x4* __restrict__ in, int lane,
                                        f32x4* __restrict__ out) {
  constexpr int T = CIN / 16;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const f32x4 wf = frags[(nt * T + t) * 64 + lane];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u], in[t][u], acc, 0, 0, 0);
    }
    out[nt] = acc;
    __builtin_amdgcn_sched_barrier(0);                   // keep the fragment reads of later tiles from being hoisted (spills)
  }
}
// the same staging for a (cin, cout)-layout matrix (1x1 convolution kernels): frag = W[16 t + 4 (lane >> 4) .. +3][16 nt + (lane & 15)]
template <int CIN, int NT>
__device__ static inline void stage_frags_t(const float* __restrict__ W, int cout, f32x4* __restrict__ dst, int tid, int nthreads) {
  constexpr int T = CIN / 16;
  for (int i = tid; i < NT * T * 64; i += nthreads) {
    const int lane = i & 63, ft = i >> 6;
    const int nt = ft / T, t = ft - nt * T;
    const float* wp = W + (int64_t)(16 * t + 4 * (lane >> 4)) * cout + 16 * nt + (lane & 15);
    dst[i] = (f32x4){wp[0], wp[cout], wp[2 * (int64_t)cout], wp[3 * (int64_t)cout]};
  }
}
constexpr int LH_WAVES = 8;
constexpr int LH_F_DW0 = 0, LH_F_DW1 = LH_F_DW0 + 6 * 4, LH_F_KW0 = LH_F_DW1 + 8 * 6, LH_F_KW1 = LH_F_KW0 + 2 * 4,
              LH_F_SW0 = LH_F_KW1 + 1 * 2, LH_F_SW1 = LH_F_SW0 + 2 * 4, LH_FRAGS = LH_F_SW1 + 1 * 2;      // 92 fragments of 1 KB
__global__ __launch_bounds__(LH_WAVES * 64) void local_heads_kernel(const LocalHeadsArgs p) {
  extern __shared__ __attribute__((aligned(16))) f32x4 lh_frags[];      // [LH_FRAGS][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g4 = lane >> 4;
  stage_frags<64, 6>(p.dw0, 96, lh_frags + LH_F_DW0 * 64, tid, LH_WAVES * 64);
  stage_frags<96, 8>(p.dw1, 128, lh_frags + LH_F_DW1 * 64, tid, LH_WAVES * 64);
  stage_frags<64, 2>(p.kw0, 32, lh_frags + LH_F_KW0 * 64, tid, LH_WAVES * 64);
  stage_frags<32, 1>(p.kw1, 3, lh_frags + LH_F_KW1 * 64, tid, LH_WAVES * 64);
  stage_frags<64, 2>(p.sw0, 32, lh_frags + LH_F_SW0 * 64, tid, LH_WAVES * 64);
  stage_frags<32, 1>(p.sw1, 1, lh_frags + LH_F_SW1 * 64, tid, LH_WAVES * 64);
  if (p.lw) stage_frags_t<64, 4>(p.lw, 64, lh_frags + LH_FRAGS * 64, tid, LH_WAVES * 64);
  __syncthreads();
  int64_t n = p.n;
  if (p.n_dev) n = min((int64_t)*p.n_dev, n);
  const int64_t ntiles = (n + 15) >> 4;
  for (int64_t tile = (int64_t)blockIdx.x * LH_WAVES + wave; tile < ntiles; tile += (int64_t)gridDim.x * LH_WAVES) {
    const int64_t row = tile * 16 + l15;
    const bool ok = row < n;
    f32x4 x[4];
    auto load4 = [&](const float* base, int t) -> f32x4 {     // four channels of the row: fp32, or bf16 widened
      if (!ok) return (f32x4){0.f, 0.f, 0.f, 0.f};
      if (p.in_bf16) {
        const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + row * 64 + 16 * t + 4 * g4);
        return (f32x4){bf2f(h.x & 0xFFFFu), bf2f(h.x >> 16), bf2f(h.y & 0xFFFFu), bf2f(h.y >> 16)};
      }
      return *reinterpret_cast<const f32x4*>(base + row * 64 + 16 * t + 4 * g4);
    };
#pragma unroll
    for (int t = 0; t < 4; ++t) x[t] = load4(p.x, t);
    if (p.lw) {
      // ---- MinkHead's last lateral (models/minkgl.py:46-60): conv1x1(x3) + the transposed convolution's output, in the
      //      accumulation order of the dense kernel it replaces (bitwise the same rows); the 64-channel map the three heads read
      //      never goes to memory
      f32x4 r[4], l[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) r[t] = load4(p.lres, t);
      mlp_layer<64, 4>(lh_frags + LH_FRAGS * 64, x, lane, l);
#pragma unroll
      for (int t = 0; t < 4; ++t) x[t] = l[t] + r[t];
    }
    // ---- descriptor decoder + L2 normalisation
    {
      f32x4 h[6], o[8];
      mlp_layer<64, 6>(lh_frags + LH_F_DW0 * 64, x, lane, h);
#pragma unroll
      for (int nt = 0; nt < 6; ++nt) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.db0 + 16 * nt + 4 * g4);
#pragma unroll
        for (int r = 0; r < 4; ++r) h[nt][r] = fmaxf(h[nt][r] + b[r], 0.f);
      }
      mlp_layer<96, 8>(lh_frags + LH_F_DW1 * 64, h, lane, o);
      float ss = 0.f;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.db1 + 16 * nt + 4 * g4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[nt][r] += b[r];
          ss += o[nt][r] * o[nt][r];
        }
      }
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);      // F.normalize(p=2, dim=1, eps=1e-12)
      if (ok) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
          *reinterpret_cast<f32x4*>(p.out_desc + row * 128 + 16 * nt + 4 * g4) = o[nt] * inv;
      }
    }
    // ---- keypoint regressor -> position of the keypoint in metres
    {
      f32x4 h[2], o[1];
      mlp_layer<64, 2>(lh_frags + LH_F_KW0 * 64, x, lane, h);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.kb0 + 16 * nt + 4 * g4);
#pragma unroll
        for (int r = 0; r < 4; ++r) h[nt][r] = fmaxf(h[nt][r] + b[r], 0.f);
      }
      mlp_layer<32, 1>(lh_frags + LH_F_KW1 * 64, h, lane, o);
      if (ok && g4 == 0) {                                   // lane (row, g = 0) holds output units 0..3
        const float ox = p.ignore_offsets ? 0.f : tanhf(o[0][0] + p.kb1[0]);
        const float oy = p.ignore_offsets ? 0.f : tanhf(o[0][1] + p.kb1[1]);
        const float oz = p.ignore_offsets ? 0.f : tanhf(o[0][2] + p.kb1[2]);
        keypoint_position(p.keys[row], p.level, p.cb, ox, oy, oz, p.mode, p.s0, p.s1, p.s2, p.out_kp + row * 3);
      }
    }
    // ---- sigma regressor
    {
      f32x4 h[2], o[1];
      mlp_layer<64, 2>(lh_frags + LH_F_SW0 * 64, x, lane, h);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.sb0 + 16 * nt + 4 * g4);
#pragma unroll
        for (int r = 0; r < 4; ++r) h[nt][r] = fmaxf(h[nt][r] + b[r], 0.f);
      }
      mlp_layer<32, 1>(lh_frags + LH_F_SW1 * 64, h, lane, o);
      if (ok && g4 == 0) p.out_sigma[row] = apply_act(o[0][0] + p.sb1[0], ACT_SOFTPLUS);
    }
  }
}
// ------------------------------------------------------------------ the three heads on the fp16 matrix pipe (round 6)
// local_heads_kernel above is bound by v_mfma_f32_16x16x4_f32: 432 MFMAs of 32 cycles per 16-row tile, two waves per SIMD —
// 1 831 tiles x 13.8 k cycles / 1 024 SIMDs = 10 us of the 36 us launch at best.  The heads' six Linear layers here run with
// the two-way fp16 split of sconv_split.hip instead (weights packed ONCE per model as hi | lo fragments in the contraction
// order of the register chain, one power-of-two scale per matrix; activations split in registers; three products on
// v_mfma_f32_16x16x32_f16, small terms first): 138 MFMAs of 16 cycles.  The lateral 1x1 convolution in front stays on the exact
// pipe in the accumulation order of the dense kernel (the fused / unfused switch stays bitwise).  Deviation of the outputs from
// the exact kernel: < 3e-6 of the largest value per output (tests/test_gpu_fusions.py); non-finite head outputs raise the range
// flag of the plan (an input beyond the fp16 range cannot pass silently).  egonn_ctx_set_exact_fp32 selects the kernel above.
typedef _Float16 lh_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 lh_f16x2 __attribute__((ext_vector_type(2)));
typedef float lh_f32x2 __attribute__((ext_vector_type(2)));
__host__ __device__ static inline int lh_chan(int g, int e) { return e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4); }
// fragments (1 KB hi + 1 KB lo each) per matrix: NT * CIN / 32
constexpr int LHS_DW0 = 0, LHS_DW1 = LHS_DW0 + 6 * 2, LHS_KW0 = LHS_DW1 + 8 * 3, LHS_KW1 = LHS_KW0 + 2 * 2, LHS_SW0 = LHS_KW1 + 1 * 1,
              LHS_SW1 = LHS_SW0 + 2 * 2, LHS_FRAGS = LHS_SW1 + 1 * 1;                                   // 46 fragment pairs = 92 KB
struct LhsMat { int cin, nout, nt, first; };
__constant__ LhsMat lhs_mats[6] = {{64, 96, 6, LHS_DW0}, {96, 128, 8, LHS_DW1}, {64, 32, 2, LHS_KW0}, {32, 3, 1, LHS_KW1},
                                   {64, 32, 2, LHS_SW0}, {32, 1, 1, LHS_SW1}};
__global__ void lhs_absmax_kernel(const float* const* __restrict__ W, uint32_t* __restrict__ trailer) {
  const int mi = blockIdx.x;
  const LhsMat M = lhs_mats[mi];
  const float* w = W[mi];
  float m = 0.f;
  for (int i = threadIdx.x; i < M.cin * M.nout; i += blockDim.x) {
    const float v = fabsf(w[i]);
    m = (v == v && v < INFINITY) ? fmaxf(m, v) : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(trailer + 8 + mi, __float_as_uint(m));
}
__device__ static inline float lhs_scale_of(uint32_t maxbits) {        // power of two s with s * max in [2^13, 2^14)
  const int e = (int)(maxbits >> 23) - 127;
  const int se = min(max(13 - e, -100), 100);
  return __uint_as_float((uint32_t)(se + 127) << 23);
}
// out[(first + nt * KB + kb) * 2 + part][lane][e] = part(s * W[16 nt + (lane & 15)][32 kb + lh_chan(lane >> 4, e)]) (fp16),
// rows beyond nout: zeros; behind the fragments: float 1/s per matrix [0..5], max bits [8..13]
__global__ void lhs_pack_kernel(const float* const* __restrict__ W, uint16_t* __restrict__ out, uint32_t* __restrict__ trailer) {
  const int mi = blockIdx.y;
  const LhsMat M = lhs_mats[mi];
  const int KB = M.cin / 32;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const float sc = lhs_scale_of(trailer[8 + mi]);
  if (t == 0) reinterpret_cast<float*>(trailer)[mi] = 1.f / sc;
  if (t >= M.nt * KB * 2 * 64 * 8) return;
  int r = t;
  const int e = r & 7; r >>= 3;
  const int lane = r & 63; r >>= 6;
  const int part = r & 1; r >>= 1;
  const int kb = r % KB, nt = r / KB;
  const int unit = 16 * nt + (lane & 15), k = 32 * kb + lh_chan(lane >> 4, e);
  const float v = unit < M.nout ? sc * W[mi][(int64_t)unit * M.cin + k] : 0.f;
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  out[(int64_t)M.first * 2 * 512 + t] = __builtin_bit_cast(uint16_t, part == 0 ? hi : lo);
}
size_t local_heads_pack_bytes() { return (size_t)LHS_FRAGS * 2048 + 64; }
int local_heads_pack(const float* const* w6_dev /*device array of the six weight pointers*/, void* out, hipStream_t stream) {
  uint32_t* trailer = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(out) + (size_t)LHS_FRAGS * 2048);
  HIP_CHECK(hipMemsetAsync(trailer, 0, 64, stream));
  hipLaunchKernelGGL(lhs_absmax_kernel, dim3(6), dim3(256), 0, stream, w6_dev, trailer);
  hipLaunchKernelGGL(lhs_pack_kernel, dim3((8 * 3 * 2 * 64 * 8 + 255) / 256, 6), dim3(256), 0, stream, w6_dev,
                     reinterpret_cast<uint16_t*>(out), trailer);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}
__device__ static inline void lhs_split8(const f32x4& a0, const f32x4& a1, lh_f16x8& hi, lh_f16x8& lo) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float x0 = q < 2 ? a0[2 * q] : a1[2 * q - 4], x1 = q < 2 ? a0[2 * q + 1] : a1[2 * q - 3];
    const lh_f16x2 h = __builtin_convertvector((lh_f32x2){x0, x1}, lh_f16x2);
    const lh_f32x2 hf = __builtin_convertvector(h, lh_f32x2);
    const lh_f16x2 l = __builtin_convertvector((lh_f32x2){x0 - hf[0], x1 - hf[1]}, lh_f16x2);
    hi[2 * q] = h[0]; hi[2 * q + 1] = h[1];
    lo[2 * q] = l[0]; lo[2 * q + 1] = l[1];
  }
}
// out[nt] = winv * sum_kb (W_lo ah + W_hi al + W_hi ah): in = f32x4 per 16 input channels (the register chain's layout)
template <int CIN, int NT>
__device__ static inline void lhs_layer(const f32x4* __restrict__ frags /* pairs: hi, lo */, const f32x4* __restrict__ in, int lane,
                                        float winv, f32x4* __restrict__ out) {
  constexpr int KB = CIN / 32;
  lh_f16x8 ah[KB], al[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) lhs_split8(in[2 * kb], in[2 * kb + 1], ah[kb], al[kb]);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const lh_f16x8 wh = __builtin_bit_cast(lh_f16x8, frags[((nt * KB + kb) * 2) * 64 + lane]);
      const lh_f16x8 wl = __builtin_bit_cast(lh_f16x8, frags[((nt * KB + kb) * 2 + 1) * 64 + lane]);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah[kb], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al[kb], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah[kb], acc, 0, 0, 0);
    }
    out[nt] = acc * winv;
    __builtin_amdgcn_sched_barrier(0);
  }
}
struct LocalHeadsSplitArgs {
  LocalHeadsArgs a;
  const void* pack;          // local_heads_pack
  int32_t* flags;            // the plan's flag word (bit 3: non-finite head output)
};
__global__ __launch_bounds__(LH_WAVES * 64) void local_heads_split_kernel(const LocalHeadsSplitArgs q) {
  const LocalHeadsArgs& p = q.a;
  extern __shared__ __attribute__((aligned(16))) f32x4 lh_frags[];      // [LHS_FRAGS * 2][64] split fragments, then the lateral's 16
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g4 = lane >> 4;
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(q.pack);
    for (int i = tid; i < LHS_FRAGS * 2 * 64; i += LH_WAVES * 64) lh_frags[i] = src[i];
  }
  f32x4* const lat = lh_frags + LHS_FRAGS * 2 * 64;
  if (p.lw) stage_frags_t<64, 4>(p.lw, 64, lat, tid, LH_WAVES * 64);
  const float* winv = reinterpret_cast<const float*>(reinterpret_cast<const char*>(q.pack) + (size_t)LHS_FRAGS * 2048);
  const float wi0 = winv[0], wi1 = winv[1], wi2 = winv[2], wi3 = winv[3], wi4 = winv[4], wi5 = winv[5];
  __syncthreads();
  int64_t n = p.n;
  if (p.n_dev) n = min((int64_t)*p.n_dev, n);
  const int64_t ntiles = (n + 15) >> 4;
  for (int64_t tile = (int64_t)blockIdx.x * LH_WAVES + wave; tile < ntiles; tile += (int64_t)gridDim.x * LH_WAVES) {
    const int64_t row = tile * 16 + l15;
    const bool ok = row < n;
    f32x4 x[4];
    auto load4 = [&](const float* base, int t) -> f32x4 {
      if (!ok) return (f32x4){0.f, 0.f, 0.f, 0.f};
      if (p.in_bf16) {
        const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + row * 64 + 16 * t + 4 * g4);
        return (f32x4){bf2f(h.x & 0xFFFFu), bf2f(h.x >> 16), bf2f(h.y & 0xFFFFu), bf2f(h.y >> 16)};
      }
      return *reinterpret_cast<const f32x4*>(base + row * 64 + 16 * t + 4 * g4);
    };
#pragma unroll
    for (int t = 0; t < 4; ++t) x[t] = load4(p.x, t);
    if (p.lw) {                                              // the lateral: exact pipe, the dense kernel's order (bitwise)
      f32x4 r[4], l[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) r[t] = load4(p.lres, t);
      mlp_layer<64, 4>(lat, x, lane, l);
#pragma unroll
      for (int t = 0; t < 4; ++t) x[t] = l[t] + r[t];
    }
    float guard = 0.f;                                       // (v - v) is NaN for a non-finite v: sticky under addition
    // ---- descriptor decoder + L2 normalisation
    {
      f32x4 h[6], o[8];
      lhs_layer<64, 6>(lh_frags + LHS_DW0 * 2 * 64, x, lane, wi0, h);
#pragma unroll
      for (int nt = 0; nt < 6; ++nt) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.db0 + 16 * nt + 4 * g4);
#pragma unroll
        for (int r = 0; r < 4; ++r) h[nt][r] = fmaxf(h[nt][r] + b[r], 0.f);
      }
      lhs_layer<96, 8>(lh_frags + LHS_DW1 * 2 * 64, h, lane, wi1, o);
      float ss = 0.f;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.db1 + 16 * nt + 4 * g4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[nt][r] += b[r];
          ss += o[nt][r] * o[nt][r];
        }
      }
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      guard += ss - ss;
      const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
      if (ok) {
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
          *reinterpret_cast<f32x4*>(p.out_desc + row * 128 + 16 * nt + 4 * g4) = o[nt] * inv;
      }
    }
    // ---- keypoint regressor -> position of the keypoint in metres
    {
      f32x4 h[2], o[1];
      lhs_layer<64, 2>(lh_frags + LHS_KW0 * 2 * 64, x, lane, wi2, h);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.kb0 + 16 * nt + 4 * g4);
#pragma unroll
        for (int r = 0; r < 4; ++r) h[nt][r] = fmaxf(h[nt][r] + b[r], 0.f);
      }
      lhs_layer<32, 1>(lh_frags + LHS_KW1 * 2 * 64, h, lane, wi3, o);
      if (ok && g4 == 0) {
        guard += (o[0][0] - o[0][0]) + (o[0][1] - o[0][1]) + (o[0][2] - o[0][2]);
        const float ox = p.ignore_offsets ? 0.f : tanhf(o[0][0] + p.kb1[0]);
        const float oy = p.ignore_offsets ? 0.f : tanhf(o[0][1] + p.kb1[1]);
        const float oz = p.ignore_offsets ? 0.f : tanhf(o[0][2] + p.kb1[2]);
        keypoint_position(p.keys[row], p.level, p.cb, ox, oy, oz, p.mode, p.s0, p.s1, p.s2, p.out_kp + row * 3);
      }
    }
    // ---- sigma regressor
    {
      f32x4 h[2], o[1];
      lhs_layer<64, 2>(lh_frags + LHS_SW0 * 2 * 64, x, lane, wi4, h);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.sb0 + 16 * nt + 4 * g4);
#pragma unroll
        for (int r = 0; r < 4; ++r) h[nt][r] = fmaxf(h[nt][r] + b[r], 0.f);
      }
      lhs_layer<32, 1>(lh_frags + LHS_SW1 * 2 * 64, h, lane, wi5, o);
      if (ok && g4 == 0) {
        guard += o[0][0] - o[0][0];
        p.out_sigma[row] = apply_act(o[0][0] + p.sb1[0], ACT_SOFTPLUS);
      }
    }
    if (ok && guard != 0.f && q.flags) atomicOr(q.flags, 8);
  }
}

int local_heads_forward(const float* x, int64_t n, const int32_t* n_dev, const float* const* w /*12 pointers*/,
                        const uint64_t* keys, int level, int cb, int mode, const float* step, int ignore_offsets,
                        float* out_desc, float* out_kp, float* out_sigma, hipStream_t stream, const float* lateral_w,
                        const float* lateral_res, int in_bf16, const void* split_pack, int32_t* flags) {
  if (n == 0) return EGONN_OK;
  EGONN_REQUIRE((lateral_w == nullptr) == (lateral_res == nullptr), EGONN_ERR_INVALID, "local heads: lateral kernel and residual go together");
  LocalHeadsArgs a;
  a.x = x; a.n = n; a.n_dev = n_dev;
  a.lw = lateral_w; a.lres = lateral_res;
  a.in_bf16 = in_bf16 ? 1 : 0;
  EGONN_REQUIRE(!in_bf16 || lateral_w, EGONN_ERR_INVALID, "local heads: bf16 input rows only through the fused lateral");
  a.dw0 = w[0]; a.db0 = w[1]; a.dw1 = w[2]; a.db1 = w[3];
  a.kw0 = w[4]; a.kb0 = w[5]; a.kw1 = w[6]; a.kb1 = w[7];
  a.sw0 = w[8]; a.sb0 = w[9]; a.sw1 = w[10]; a.sb1 = w[11];
  a.keys = keys; a.level = level; a.cb = cb; a.mode = mode; a.ignore_offsets = ignore_offsets;
  a.s0 = step[0]; a.s1 = mode ? step[1] : step[0]; a.s2 = mode ? step[2] : step[0];
  a.out_desc = out_desc; a.out_kp = out_kp; a.out_sigma = out_sigma;
  const int64_t tiles = cdiv(n, 16);
  const size_t lds = (size_t)(LH_FRAGS + (lateral_w ? 16 : 0)) * 64 * sizeof(f32x4);
  static AttrOnce attr_done;
  if (attr_done.need()) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&local_heads_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done.mark(); 
  }
  const unsigned grid = (unsigned)std::min<int64_t>(cdiv(tiles, LH_WAVES), 256);      // one workgroup per CU holds the weights
  if (split_pack) {
    static AttrOnce attr2;
    if (attr2.need()) {
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&local_heads_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr2.mark();
    }
    LocalHeadsSplitArgs q;
    q.a = a; q.pack = split_pack; q.flags = flags;
    const size_t lds2 = (size_t)(LHS_FRAGS * 2 + (lateral_w ? 16 : 0)) * 64 * sizeof(f32x4);
    hipLaunchKernelGGL(local_heads_split_kernel, dim3(grid), dim3(LH_WAVES * 64), lds2, stream, q);
    HIP_CHECK(hipGetLastError());
    return EGONN_OK;
  }
  hipLaunchKernelGGL(local_heads_kernel, dim3(grid), dim3(LH_WAVES * 64), lds, stream, a);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------ top-k (smallest sigma, ascending, ties by row)
// MinkLocGLEvaluator.get_keypoints_idxes (eval/evaluate.py:352-361: torch.topk(sigma, n_k, largest=False) per scan) +
// the gather of the selected keypoints / descriptors, ONE launch: workgroup = scan.  Keys = (monotone sigma bits << 32 |
// row inside the scan) are unique, so the order is total (ties by Z-order row, as the stable radix sort it replaces).
// Radix SELECT, then a sort of the k winners only: 11-bit digits of the 64-bit key from the top, one LDS histogram per
// digit over the rows that still match the prefix, until the tie group is taken whole (two or three passes on real
// saliencies, six at most: the keys are unique) -> a threshold key; the k keys at or below it are compacted into LDS and
// ordered by counting (k <= 512) or by a bitonic network.  (The r01/r02 kernel sorted the whole scan through a 2048-key
// bitonic network, 66 barrier stages per 1920 rows: 49 us at 23 k rows per scan.)
__device__ static inline uint32_t sigma_bits(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // monotone float -> uint
}
constexpr int TK_THREADS = 1024, TK_BINS = 2048, TK_CACHE = 8;
__global__ __launch_bounds__(TK_THREADS) void select_topk_kernel(const float* __restrict__ sigma, const int32_t* __restrict__ boff,
                                                                 int k, int S, const float* __restrict__ kp,
                                                                 const float* __restrict__ desc, int dc,
                                                                 int32_t* __restrict__ sel_rows, int32_t* __restrict__ sel_count,
                                                                 float* __restrict__ out_kp, float* __restrict__ out_desc) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long skeys[];   // [S] winners, S = pow2 >= k; + [k] when ranked
  __shared__ uint32_t s_hist[TK_BINS];
  __shared__ uint32_t s_wave[TK_THREADS / 64];
  __shared__ uint32_t s_pick[3];                         // digit, rows below it, rows in it
  __shared__ uint32_t s_n;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int32_t r0 = boff[b], nb = boff[b + 1] - r0;
  const int kk = min(k, nb);
  // the scan's keys: the first TK_CACHE * 1024 rows stay in registers, longer scans re-read the tail
  uint32_t sb[TK_CACHE];
#pragma unroll
  for (int i = 0; i < TK_CACHE; ++i) {
    const int r = i * TK_THREADS + tid;
    sb[i] = r < nb ? sigma_bits(sigma[r0 + r]) : 0xFFFFFFFFu;
  }
  auto key_of = [&](uint32_t bits, int r) { return ((unsigned long long)bits << 32) | (unsigned)r; };
  unsigned long long T = 0;                              // threshold: the kk smallest keys are the keys <= T
  if (kk > 0) {
    unsigned long long prefix = 0;
    uint32_t need = (uint32_t)kk;
    const int shifts[6] = {53, 42, 31, 20, 9, 0};
#pragma unroll 1
    for (int p = 0; p < 6; ++p) {
      const int sh = shifts[p], bits = p == 5 ? 9 : 11;
      const uint32_t dmask = (1u << bits) - 1;
      for (int i = tid; i < TK_BINS; i += TK_THREADS) s_hist[i] = 0;
      __syncthreads();
      auto count = [&](uint32_t sbits, int r) {
        const unsigned long long key = key_of(sbits, r);
        if (p == 0 || (key >> (sh + bits)) == prefix) atomicAdd(&s_hist[(uint32_t)(key >> sh) & dmask], 1u);
      };
#pragma unroll
      for (int i = 0; i < TK_CACHE; ++i) {
        const int r = i * TK_THREADS + tid;
        if (r < nb) count(sb[i], r);
      }
      for (int r = TK_CACHE * TK_THREADS + tid; r < nb; r += TK_THREADS) count(sigma_bits(sigma[r0 + r]), r);
      __syncthreads();
      // digit that holds the need-th smallest key: exclusive scan of the histogram, two bins per thread
      const uint32_t h0 = s_hist[2 * tid], h1 = s_hist[2 * tid + 1];
      uint32_t inc = h0 + h1;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
      }
      if (lane == 63) s_wave[wave] = inc;
      __syncthreads();
      uint32_t ex = inc - (h0 + h1);
      for (int w = 0; w < wave; ++w) ex += s_wave[w];
      if (ex < need && need <= ex + h0) {
        s_pick[0] = 2 * tid; s_pick[1] = ex; s_pick[2] = h0;
      } else if (ex + h0 < need && need <= ex + h0 + h1) {
        s_pick[0] = 2 * tid + 1; s_pick[1] = ex + h0; s_pick[2] = h1;
      }
      __syncthreads();
      const uint32_t digit = s_pick[0], below = s_pick[1], inbin = s_pick[2];
      prefix = (prefix << bits) | digit;
      need -= below;
      if (inbin == need || sh == 0) {                    // the tie group is taken whole
        T = sh ? ((prefix << sh) | ((1ull << sh) - 1)) : prefix;
        break;
      }
    }
  }
  // ---- the kk winners -> LDS (any order), then ordered
  if (tid == 0) s_n = 0;
  for (int i = tid; i < S; i += TK_THREADS) skeys[i] = ~0ull;
  __syncthreads();
  if (kk > 0) {
    auto take = [&](uint32_t sbits, int r) {
      const unsigned long long key = key_of(sbits, r);
      if (key <= T) skeys[atomicAdd(&s_n, 1u)] = key;
    };
#pragma unroll
    for (int i = 0; i < TK_CACHE; ++i) {
      const int r = i * TK_THREADS + tid;
      if (r < nb) take(sb[i], r);
    }
    for (int r = TK_CACHE * TK_THREADS + tid; r < nb; r += TK_THREADS) take(sigma_bits(sigma[r0 + r]), r);
  }
  __syncthreads();
  if (k <= 512) {                                        // order by counting: every key reads all the others (broadcasts)
    unsigned long long* sorted = skeys + S;
    if (tid < kk) {
      const unsigned long long mine = skeys[tid];
      int rank = 0;
      for (int j = 0; j < kk; ++j) rank += skeys[j] < mine ? 1 : 0;
      sorted[rank] = mine;
    }
    __syncthreads();
    if (tid < kk) skeys[tid] = sorted[tid];
    __syncthreads();
  } else {
    for (int k2 = 2; k2 <= S; k2 <<= 1)
      for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
        for (int e = tid; e < S / 2; e += TK_THREADS) {
          const int i = ((e & ~(j2 - 1)) << 1) | (e & (j2 - 1));
          const int q = i | j2;
          const bool up = (i & k2) == 0;
          const unsigned long long x = skeys[i], y = skeys[q];
          if ((x > y) == up) { skeys[i] = y; skeys[q] = x; }
        }
        __syncthreads();
      }
  }
  // ---- selected rows + gathered keypoints / descriptors (padded with -1 / zeros)
  for (int i = tid; i < k; i += TK_THREADS) sel_rows[(int64_t)b * k + i] = (i < kk) ? r0 + (int32_t)(skeys[i] & 0xFFFFFFFFu) : -1;
  if (tid == 0) sel_count[b] = kk;
  if (out_desc) {
    if ((dc & 3) == 0) {                                 // 16-byte pieces of the descriptor rows
      const int q4 = dc >> 2;
      for (int t = tid; t < k * q4; t += TK_THREADS) {
        const int i = t / q4, c = t - i * q4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < kk) v = *reinterpret_cast<const float4*>(desc + (int64_t)(r0 + (int32_t)(skeys[i] & 0xFFFFFFFFu)) * dc + 4 * c);
        *reinterpret_cast<float4*>(out_desc + ((int64_t)b * k + i) * dc + 4 * c) = v;
      }
    } else {
      for (int64_t t = tid; t < (int64_t)k * dc; t += TK_THREADS) {
        const int i = (int)(t / dc), c = (int)(t - (int64_t)i * dc);
        out_desc[((int64_t)b * k + i) * dc + c] = (i < kk) ? desc[(int64_t)(r0 + (int32_t)(skeys[i] & 0xFFFFFFFFu)) * dc + c] : 0.f;
      }
    }
    for (int t = tid; t < k * 3; t += TK_THREADS) {
      const int i = t / 3, c = t - i * 3;
      out_kp[((int64_t)b * k + i) * 3 + c] = (i < kk) ? kp[(int64_t)(r0 + (int32_t)(skeys[i] & 0xFFFFFFFFu)) * 3 + c] : 0.f;
    }
  }
}
int select_topk(const float* sigma, const int32_t* boff_dev, int B, int k, const float* kp, const float* desc, int dc,
                int32_t* sel_rows, int32_t* sel_count, float* out_kp, float* out_desc, hipStream_t stream) {
  EGONN_REQUIRE(k >= 1 && k <= 8192, EGONN_ERR_INVALID, "select_keypoints: n_k=%d outside [1, 8192]", k);
  int S = 64;
  while (S < k) S <<= 1;
  const size_t lds = (size_t)(S + (k <= 512 ? k : 0)) * sizeof(unsigned long long);
  static AttrOnce attr_done;
  if (attr_done.need()) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&select_topk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  128 * 1024));
    attr_done.mark(); 
  }
  hipLaunchKernelGGL(select_topk_kernel, dim3(B), dim3(TK_THREADS), lds, stream, sigma, boff_dev, k, S, kp, desc, dc, sel_rows,
                     sel_count, out_kp, out_desc);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

}  // namespace egonn
