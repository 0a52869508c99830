// First-layer convolution (k=5, Cin=1) and the plain reference kernel for channel plans the MFMA kernel does not cover.
// The sparse-convolution kernel proper lives in sconv.hip.
//
// conv0_k5_kernel — the 5x5x5, Cin=1 first layer (models/minkgl.py:100 `convs[0]`): pure lookup work.  One wave per 16
// level-0 rows, the 27 adjacent 4x4x4 blocks' occupancy masks sit in LDS, the 125 offsets are bit tests spread over the
// lanes that feed v_mfma_f32_16x16x4_f32 directly.  No kernel map is materialised.
#include <algorithm>
#include <type_traits>
#include <utility>

#include "common.h"
#include <hip/hip_ext.h>
#include "kernels.h"

namespace egonn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ naive reference kernel (bring-up / cross-check)
__global__ void sconv_naive_kernel(const float* __restrict__ in, const int32_t* __restrict__ nbr,
                                   const float* __restrict__ W, const float* __restrict__ scale,
                                   const float* __restrict__ shift, int relu, float* __restrict__ out, int32_t n_out,
                                   int K, int cin, int cout) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)n_out * cout) return;
  const int32_t o = (int32_t)(t / cout);
  const int co = (int)(t - (int64_t)o * cout);
  float acc = 0.f;
  for (int k = 0; k < K; ++k) {
    const int32_t j = nbr[(int64_t)o * K + k];
    if (j < 0) continue;
    const float* f = in + (int64_t)j * cin;
    const float* w = W + ((int64_t)k * cin) * cout + co;
    float part = 0.f;
    for (int ci = 0; ci < cin; ++ci) part = fmaf(f[ci], w[(int64_t)ci * cout], part);
    acc += part;
  }
  if (scale) acc = acc * scale[co] + shift[co];
  if (relu) acc = fmaxf(acc, 0.f);
  out[t] = acc;
}

int sconv_naive(const float* in, const int32_t* nbr, const float* W, const float* scale, const float* shift, int relu,
                float* out, int64_t n_out, int K, int cin, int cout, hipStream_t stream) {
  if (n_out == 0) return EGONN_OK;
  const int64_t total = n_out * cout;
  hipLaunchKernelGGL(sconv_naive_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, in, nbr, W, scale,
                     shift, relu, out, (int32_t)n_out, K, cin, cout);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------ conv0: k=5, Cin=1 -> COUT0 channels
// out[v][c] = sum_k f[nbr_k(v)] * W[k][c]  ==  A[16 voxels][128 (125 offsets, zero padded)] @ W[128][32]
// One wave owns 16 consecutive level-0 rows.  Every lane builds exactly the 32 A-operands it feeds to
// v_mfma_f32_16x16x4_f32 itself (row = lane & 15, offsets k = 16t + 4(lane>>4) + s), each one an LDS
// occupancy-mask test + popcount on the 27 blocks around the voxel's 4x4x4 block (block adjacency = the k=3
// table of level 2), so neither a kernel map nor an LDS A-tile is materialised.  W lives in 64 B-fragment
// registers for the whole (persistent) wave.
static constexpr int COUT0 = 32;

// UNIT: the (N,1) input features are all ones (what the reference always feeds: eval/evaluate.py:334,
// datasets/dataset_utils.py:80) -> an A-operand is just the occupancy bit, no rank/popcount, no feature gather.
template <bool UNIT, bool OUT_BF16>
__global__ __launch_bounds__(256) void conv0_k5_kernel(const float* __restrict__ feat,         // [n0] (Cin = 1)
                                                        const uint64_t* __restrict__ vkeys,     // level 0
                                                        const int32_t* __restrict__ g0,         // level-2 block of row
                                                        const uint64_t* __restrict__ t2m,       // [n2][27] masks
                                                        const int32_t* __restrict__ t2s,        // [n2][27] first rows
                                                        const int32_t* __restrict__ counts,     // device rows per level
                                                        int32_t cap2, int32_t cap0,             // capacities (grid sizing)
                                                        const float* __restrict__ W,            // [125][32]
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int relu,
                                                        void* __restrict__ out_v,
                                                        const uint16_t* __restrict__ lut) {
  // 27 neighbour blocks per row + one all-zero mask (slot 27: where the LUT sends the 3 padding offsets, so the lookup has
  // no compare/select); row stride 29 keeps the 16 rows of a tile on distinct banks
  constexpr int MS = 29;
  __shared__ uint64_t s_m[4][16][MS];
  __shared__ int32_t s_s[4][16][27];
  // (local voxel position inside its 4x4x4 block, kernel offset) -> (adjacent-block slot << 6 | bit inside that
  // block's occupancy mask); the 3 padding offsets point at the zero mask.  Built once per (persistent) workgroup.
  // rows padded to 132 entries: the 16 rows of a tile read the same column of 16 different LUT rows, which a 256-byte
  // row stride put on one bank (16-way conflict on every lookup)
  constexpr int LUT_STRIDE = 132;
  __shared__ __attribute__((aligned(8))) uint16_t s_lut[64 * LUT_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g4 = lane >> 4;
  const int32_t nvox = min(__builtin_amdgcn_readfirstlane(counts[0]), cap0);
  const int32_t n2 = min(__builtin_amdgcn_readfirstlane(counts[2]), cap2);
  const int32_t ntiles = (nvox + 15) >> 4;
  for (int e = tid; e < 64 * 128 / 2; e += 256) {     // precomputed table (conv0_lut_host), 16 KB, coalesced
    const int r = e >> 6, c = e & 63;
    reinterpret_cast<uint32_t*>(s_lut)[r * (LUT_STRIDE / 2) + c] = reinterpret_cast<const uint32_t*>(lut)[e];
  }
  if (tid < 64) s_m[tid >> 4][tid & 15][27] = 0ull;
  float breg[2][32];
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int k = 16 * (q >> 2) + 4 * g4 + (q & 3);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) breg[nt][q] = (k < 125) ? W[k * COUT0 + nt * 16 + l15] : 0.f;
  }
  f32x4 sc[2], sh[2];                                    // lane (row, g) stores output channels nt*16 + 4g .. +3 of its row
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    sc[nt] = scale ? *reinterpret_cast<const f32x4*>(scale + nt * 16 + 4 * g4) : (f32x4){1.f, 1.f, 1.f, 1.f};
    sh[nt] = scale ? *reinterpret_cast<const f32x4*>(shift + nt * 16 + 4 * g4) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  // Software pipeline over the wave's tiles: while tile t is computed, the 27-neighbourhood (mask, first row) of
  // the rows of tile t+1 and the block index / low key bits of tile t+2 are in flight.  The neighbourhood comes from
  // the per-block table built at plan time (blk27_kernel), so the dependent chain is row -> block -> table entry
  // (it used to be row -> parent -> parent -> adjacency -> mask: 4 serial L2 round trips per tile).  Rows past the
  // end read block -1 -> out-of-range buffer loads -> 0 masks: no branches.
  const __amdgpu_buffer_rsrc_t m_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(t2m), 0, (int)((uint32_t)n2 * 27u * 8u), 0x00020000);
  const __amdgpu_buffer_rsrc_t s_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(t2s), 0, (int)((uint32_t)n2 * 27u * 4u), 0x00020000);
  const int tstep = gridDim.x * 4;
  constexpr int NE = (16 * 27 + 63) / 64;                    // table entries per lane
  uint64_t pm[NE];
  int32_t ps[NE];
  // rows past the end read 0 through the bounds-checked resources (block 0 / position 0): their results are
  // computed on garbage and never stored — cheaper than a branch, which would make hipcc drain vmcnt(0)
  const __amdgpu_buffer_rsrc_t g_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(g0), 0, (int)((uint32_t)nvox * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t k_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(vkeys), 0, (int)((uint32_t)nvox * 8u), 0x00020000);
  auto row_info = [&](int32_t tile, int32_t& g, uint32_t& lkbits) {
    const uint32_t r = (uint32_t)tile * 16u + (uint32_t)l15;
    g = __builtin_amdgcn_raw_buffer_load_b32(g_rsrc, (int)(r * 4u), 0, 0);
    lkbits = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(k_rsrc, (int)(r * 8u), 0, 0) & 63u;
  };
  auto table_issue = [&](int32_t g_of_lane) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int idx = lane + 64 * e;
      const int row = idx / 27, slot = idx - row * 27;
      const int32_t g = __shfl(g_of_lane, row & 15, 64);          // lane `row` (g4 = 0) holds that row's block
      const uint32_t ent = (idx < 16 * 27) ? (uint32_t)g * 27u + (uint32_t)slot : 0x3FFFFFFFu;
      const auto m2 = __builtin_amdgcn_raw_buffer_load_b64(m_rsrc, (int)(ent * 8u), 0, 0);
      pm[e] = ((uint64_t)m2[1] << 32) | (uint64_t)m2[0];
      ps[e] = __builtin_amdgcn_raw_buffer_load_b32(s_rsrc, (int)(ent * 4u), 0, 0);
    }
  };
  int32_t tile = blockIdx.x * 4 + wave;
  int32_t g_cur, g_nxt;
  uint32_t lk_cur, lk_nxt, lk_nn;
  int32_t g_nn;
  row_info(tile, g_cur, lk_cur);
  row_info(tile + tstep, g_nxt, lk_nxt);
  table_issue(g_cur);
  for (; tile < ntiles; tile += tstep) {
    const int32_t r0 = tile * 16;
    // ---- commit the prefetched neighbourhood of this tile to the wave's LDS table
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int idx = lane + 64 * e;
      if (idx < 16 * 27) {
        const int row = idx / 27, slot = idx - row * 27;
        s_m[wave][row][slot] = pm[e];
        (&s_s[wave][0][0])[idx] = ps[e];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- prefetch: table of the next tile, block index + key bits of the one after
    table_issue(g_nxt);
    row_info(tile + 2 * tstep, g_nn, lk_nn);
    const uint32_t lk = lk_cur;
    const uint16_t* lrow = s_lut + lk * LUT_STRIDE + 4 * g4;
    uint2 ent4[8];                                           // the lane's 32 entries: 8 x 4 consecutive ones (one 8-byte read each)
#pragma unroll
    for (int j = 0; j < 8; ++j) ent4[j] = *reinterpret_cast<const uint2*>(lrow + 16 * j);
    float a[32];
    const char* mrow = reinterpret_cast<const char*>(&s_m[wave][l15][0]);
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const uint32_t pair = (q & 2) ? ent4[q >> 2].y : ent4[q >> 2].x;
      const uint32_t en = (q & 1) ? (pair >> 16) : (pair & 0xFFFFu);       // (byte offset of the block's mask << 6) | bit
      const uint64_t m = *reinterpret_cast<const uint64_t*>(mrow + (en >> 6));      // rows beyond nvox have all-zero masks
      const uint32_t hit = (uint32_t)(m >> (en & 63)) & 1u;
      float v = (float)hit;
      if constexpr (!UNIT) {
        v = 0.f;
        if (hit) v = feat[s_s[wave][l15][en >> 9] + __popcll(m & ((1ull << (en & 63)) - 1))];
      }
      a[q] = v;
    }
    // operands swapped (D^T = W^T A^T): lane (row = l15, g) ends up with four consecutive output channels of its row
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[0][q], a[q], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[1][q], a[q], acc[1], 0, 0, 0);
    }
    const int32_t orow = r0 + l15;
    if (orow < nvox) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        f32x4 o = acc[nt] * sc[nt] + sh[nt];
        if (relu) {
#pragma unroll
          for (int u = 0; u < 4; ++u) o[u] = fmaxf(o[u], 0.f);
        }
        if constexpr (OUT_BF16) {
          uint32_t w[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            uint32_t x = __float_as_uint(o[u]);
            x += 0x7FFFu + ((x >> 16) & 1u);
            w[u] = x >> 16;
          }
          *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out_v) + (int64_t)orow * COUT0 + nt * 16 + 4 * g4) =
              make_uint2(w[0] | (w[1] << 16), w[2] | (w[3] << 16));
        } else {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out_v) + (int64_t)orow * COUT0 + nt * 16 + 4 * g4) = o;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    g_cur = g_nxt; lk_cur = lk_nxt;
    g_nxt = g_nn; lk_nxt = lk_nn;
  }
}

// Unit input features (what the reference always feeds): the A operand is an occupancy bit, i.e. exactly 0 or 1 — exactly
// representable in bf16 — and an fp32 weight is exactly the sum of three bf16 numbers (24 = 3 x 8 significand bits).  So the
// first layer runs on v_mfma_f32_16x16x32_bf16 with the kernel split hi + mid + lo: 2 x 4 x 3 = 24 MFMAs of 16 cycles per
// 16-row tile instead of 64 MFMAs of 32 cycles, every product exact, fp32 accumulation — same accuracy as the fp32-MFMA
// kernel (a different summation order), 5.3 x less matrix-pipe time.  Lane (row, g) supplies 8 CONSECUTIVE offsets per
// MFMA (k = 32 j + 8 g + e): its 8 table entries are one 16-byte LDS read.  Everything else (per-block neighbourhood table,
// software pipeline over the tiles, zero-mask slot for the padding offsets) is the fp32 kernel's.
typedef short bf16x8_c0 __attribute__((ext_vector_type(8)));
__device__ static inline uint32_t bf16_rne(float a) {
  uint32_t u = __float_as_uint(a);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void conv0_k5_unit_kernel(const uint64_t* __restrict__ vkeys,     // level 0
                                                             const int32_t* __restrict__ g0,         // level-2 block of row
                                                             const uint64_t* __restrict__ t2m,       // [n2][27] masks
                                                             const int32_t* __restrict__ counts, int32_t cap2, int32_t cap0,
                                                             const float* __restrict__ W,            // [125][32]
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             int relu, void* __restrict__ out_v,
                                                             const uint16_t* __restrict__ lut,
                                                             const uint4* __restrict__ wpk) {   // conv0_pack_unit (nullable)
  constexpr int MS = 29;                                   // 27 neighbour blocks + the all-zero mask (slot 27)
  constexpr int LUT_STRIDE = 136;                          // entries per row: 16-byte aligned rows, spread over the banks
  __shared__ uint64_t s_m[4][16][MS];
  __shared__ __attribute__((aligned(16))) uint16_t s_lut[64 * LUT_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g4 = lane >> 4;
  const int32_t nvox = min(__builtin_amdgcn_readfirstlane(counts[0]), cap0);
  const int32_t n2 = min(__builtin_amdgcn_readfirstlane(counts[2]), cap2);
  const int32_t ntiles = (nvox + 15) >> 4;
  for (int e = tid; e < 64 * 128 / 2; e += 256) {
    const int r = e >> 6, c = e & 63;
    reinterpret_cast<uint32_t*>(s_lut)[r * (LUT_STRIDE / 2) + c] = reinterpret_cast<const uint32_t*>(lut)[e];
  }
  if (tid < 64) s_m[tid >> 4][tid & 15][27] = 0ull;
  // W fragments: wf[nt][j][split] = bf16x8 of W[k = 32 j + 8 g + e][nt * 16 + l15], split = hi / mid / lo
  // (building them here costs a workgroup ~8 k cycles — as much as its tiles: the launch time grew linearly with the grid,
  //  profiles/r03l_conv0.txt; the model packs them once, conv0_pack_unit, and a workgroup then loads 24 x 16 bytes per lane)
  bf16x8_c0 wf[2][4][3];
  if (wpk) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) wf[nt][j][sp] = __builtin_bit_cast(bf16x8_c0, wpk[((nt * 4 + j) * 3 + sp) * 64 + lane]);
  } else
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t h[3][8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 32 * j + 8 * g4 + e;
        float w = (k < 125) ? W[k * COUT0 + nt * 16 + l15] : 0.f;
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) {
          const uint32_t b = bf16_rne(w);
          h[sp][e] = b;
          w -= __uint_as_float(b << 16);                   // exact: the residual has at most 16 (then 8) significant bits
        }
      }
#pragma unroll
      for (int sp = 0; sp < 3; ++sp) {
        const uint4 pk = make_uint4(h[sp][0] | (h[sp][1] << 16), h[sp][2] | (h[sp][3] << 16), h[sp][4] | (h[sp][5] << 16),
                                    h[sp][6] | (h[sp][7] << 16));
        wf[nt][j][sp] = __builtin_bit_cast(bf16x8_c0, pk);
      }
    }
  f32x4 sc[2], sh[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    sc[nt] = scale ? *reinterpret_cast<const f32x4*>(scale + nt * 16 + 4 * g4) : (f32x4){1.f, 1.f, 1.f, 1.f};
    sh[nt] = scale ? *reinterpret_cast<const f32x4*>(shift + nt * 16 + 4 * g4) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t m_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(t2m), 0, (int)((uint32_t)n2 * 27u * 8u), 0x00020000);
  const __amdgpu_buffer_rsrc_t g_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(g0), 0, (int)((uint32_t)nvox * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t k_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(vkeys), 0, (int)((uint32_t)nvox * 8u), 0x00020000);
  const int tstep = gridDim.x * 4;
  constexpr int NE = (16 * 27 + 63) / 64;
  uint64_t pm[NE];
  auto row_info = [&](int32_t tile, int32_t& g, uint32_t& lkbits) {
    const uint32_t r = (uint32_t)tile * 16u + (uint32_t)l15;
    g = __builtin_amdgcn_raw_buffer_load_b32(g_rsrc, (int)(r * 4u), 0, 0);
    lkbits = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(k_rsrc, (int)(r * 8u), 0, 0) & 63u;
  };
  auto table_issue = [&](int32_t g_of_lane) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int idx = lane + 64 * e;
      const int row = idx / 27, slot = idx - row * 27;
      const int32_t g = __shfl(g_of_lane, row & 15, 64);
      const uint32_t ent = (idx < 16 * 27) ? (uint32_t)g * 27u + (uint32_t)slot : 0x3FFFFFFFu;
      const auto m2 = __builtin_amdgcn_raw_buffer_load_b64(m_rsrc, (int)(ent * 8u), 0, 0);
      pm[e] = ((uint64_t)m2[1] << 32) | (uint64_t)m2[0];
    }
  };
  int32_t tile = blockIdx.x * 4 + wave;
  int32_t g_cur, g_nxt, g_nn;
  uint32_t lk_cur, lk_nxt, lk_nn;
  row_info(tile, g_cur, lk_cur);
  row_info(tile + tstep, g_nxt, lk_nxt);
  table_issue(g_cur);
  for (; tile < ntiles; tile += tstep) {
    const int32_t r0 = tile * 16;
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
    table_issue(g_nxt);
    row_info(tile + 2 * tstep, g_nn, lk_nn);
    const uint16_t* lrow = s_lut + lk_cur * LUT_STRIDE + 8 * g4;
    const char* mrow = reinterpret_cast<const char*>(&s_m[wave][l15][0]);
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint4 ent = *reinterpret_cast<const uint4*>(lrow + 32 * j);     // 8 entries: offsets 32 j + 8 g .. + 7
      const uint32_t ew[4] = {ent.x, ent.y, ent.z, ent.w};
      uint32_t pk[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t e0 = ew[q] & 0xFFFFu, e1 = ew[q] >> 16;
        const uint64_t m0 = *reinterpret_cast<const uint64_t*>(mrow + (e0 >> 6));
        const uint64_t m1 = *reinterpret_cast<const uint64_t*>(mrow + (e1 >> 6));
        const uint32_t h0 = (uint32_t)(m0 >> (e0 & 63)) & 1u, h1 = (uint32_t)(m1 >> (e1 & 63)) & 1u;
        pk[q] = h0 * 0x3F80u + h1 * 0x3F800000u;             // bf16 1.0 in the low / high half
      }
      const bf16x8_c0 av = __builtin_bit_cast(bf16x8_c0, make_uint4(pk[0], pk[1], pk[2], pk[3]));
#pragma unroll
      for (int sp = 0; sp < 3; ++sp)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][j][sp], av, acc[nt], 0, 0, 0);
    }
    const int32_t orow = r0 + l15;
    if (orow < nvox) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        f32x4 o = acc[nt] * sc[nt] + sh[nt];
        if (relu) {
#pragma unroll
          for (int u = 0; u < 4; ++u) o[u] = fmaxf(o[u], 0.f);
        }
        if constexpr (OUT_BF16) {
          *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(out_v) + (int64_t)orow * COUT0 + nt * 16 + 4 * g4) =
              make_uint2(bf16_rne(o[0]) | (bf16_rne(o[1]) << 16), bf16_rne(o[2]) | (bf16_rne(o[3]) << 16));
        } else {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out_v) + (int64_t)orow * COUT0 + nt * 16 + 4 * g4) = o;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    g_cur = g_nxt; lk_cur = lk_nxt;
    g_nxt = g_nn; lk_nxt = lk_nn;
  }
}

// W[125][32] -> the unit kernel's MFMA fragments, wpk[((nt * 4 + j) * 3 + split) * 64 + lane] = bf16x8 of
// split(W[k = 32 j + 8 (lane >> 4) + e][16 nt + (lane & 15)]), e = 0..7 — exactly what the kernel builds when it gets none
__global__ void conv0_pack_unit_kernel(const float* __restrict__ W, uint4* __restrict__ wpk) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * 4 * 64) return;
  const int lane = t & 63, j = (t >> 6) & 3, nt = t >> 8;
  const int l15 = lane & 15, g4 = lane >> 4;
  uint32_t h[3][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 32 * j + 8 * g4 + e;
    float w = (k < 125) ? W[k * COUT0 + nt * 16 + l15] : 0.f;
#pragma unroll
    for (int sp = 0; sp < 3; ++sp) {
      const uint32_t b = bf16_rne(w);
      h[sp][e] = b;
      w -= __uint_as_float(b << 16);
    }
  }
#pragma unroll
  for (int sp = 0; sp < 3; ++sp)
    wpk[((nt * 4 + j) * 3 + sp) * 64 + lane] = make_uint4(h[sp][0] | (h[sp][1] << 16), h[sp][2] | (h[sp][3] << 16),
                                                          h[sp][4] | (h[sp][5] << 16), h[sp][6] | (h[sp][7] << 16));
}
int conv0_pack_unit(const float* W, void* wpk, hipStream_t stream) {
  hipLaunchKernelGGL(conv0_pack_unit_kernel, dim3(2), dim3(256), 0, stream, W, reinterpret_cast<uint4*>(wpk));
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// (local voxel position, kernel offset) -> (8 * adjacent-block slot) << 6 | bit in that block's mask
static void conv0_lut_host(uint16_t* lut) {
  for (int lk = 0; lk < 64; ++lk)
    for (int k = 0; k < 128; ++k) {
      uint32_t v = (uint32_t)(27 * 8) << 6;              // padding offsets: the all-zero mask, bit 0
      if (k < 125) {
        const int lx = (lk & 1) | ((lk >> 2) & 2), ly = ((lk >> 1) & 1) | ((lk >> 3) & 2), lz = ((lk >> 2) & 1) | ((lk >> 4) & 2);
        const int nx = lx + k % 5 - 2, ny = ly + (k / 5) % 5 - 2, nz = lz + k / 25 - 2;
        const int slot = ((nx < 0) ? 0 : (nx > 3 ? 2 : 1)) + 3 * ((ny < 0) ? 0 : (ny > 3 ? 2 : 1)) +
                         9 * ((nz < 0) ? 0 : (nz > 3 ? 2 : 1));
        const uint32_t ux = (uint32_t)nx & 3, uy = (uint32_t)ny & 3, uz = (uint32_t)nz & 3;
        const uint32_t bit = (ux & 1) | ((uy & 1) << 1) | ((uz & 1) << 2) | ((ux & 2) << 2) | ((uy & 2) << 3) | ((uz & 2) << 4);
        v = ((uint32_t)(slot * 8) << 6) | bit;         // byte offset of the block's mask inside the row's LDS table
      }
      lut[lk * 128 + k] = (uint16_t)v;
    }
}

int conv0_lut_init(Ctx* ctx) {
  if (ctx->conv0_lut) return EGONN_OK;
  std::vector<uint16_t> h(64 * 128);
  conv0_lut_host(h.data());
  HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ctx->conv0_lut), h.size() * sizeof(uint16_t)));
  HIP_CHECK(hipMemcpy(ctx->conv0_lut, h.data(), h.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  return EGONN_OK;
}

int conv0_k5_forward(Ctx* ctx, const float* feat, const float* W, int cout, const float* scale,
                     const float* shift, int relu, void* out, int out_bf16, hipStream_t stream, const void* wpk) {
  const Plan& P = ctx->plan;
  EGONN_TRY(conv0_lut_init(ctx));
  EGONN_REQUIRE(cout == COUT0, EGONN_ERR_INVALID, "conv0: %d output channels not supported (expected %d)", cout, COUT0);
  const Level& V = P.lv[0];
  const Level& B = P.lv[2];
  if (P.cap[0] == 0) return EGONN_OK;
  const int32_t ntiles = (int32_t)cdiv(P.cap[0], 16);
  // two workgroups per CU, each wave walks ~11 tiles (batch 16): the per-workgroup set-up (16 KB lookup table -> LDS, W
  // fragments, BN vectors, first tables) is paid 512 times instead of 1 536 and every CU gets the same share.  Measured
  // (profiles/r03l_conv0.txt): 384 / 512 / 640 / 768 / 1024 / 1536 / 2560 workgroups = 39.8 / 32.8 / 42.7 / 41.0 / 37.0 / 40.3 / 45.9 us
  const unsigned grid = (unsigned)std::min<int64_t>(cdiv(ntiles, 4), 512);
  EGONN_REQUIRE(P.g0 && P.t2m && P.t2s, EGONN_ERR_STATE, "conv0: plan has no block-neighbourhood table");
#define EGONN_CONV0_LAUNCH(U, OB)                                                                                      \
  hipLaunchKernelGGL((conv0_k5_kernel<U, OB>), dim3(grid), dim3(256), 0, stream, feat, V.keys, P.g0, P.t2m, P.t2s,       \
                     ctx->dev_counts, (int32_t)P.cap[2], (int32_t)P.cap[0], W, scale, shift, relu, out, ctx->conv0_lut)
  if (feat) {
    if (out_bf16) EGONN_CONV0_LAUNCH(false, true); else EGONN_CONV0_LAUNCH(false, false);
  } else if (ctx->conv_variant == 3) {                     // cross-check path (egonn_debug_set_naive_conv): fp32-MFMA kernel
    if (out_bf16) EGONN_CONV0_LAUNCH(true, true); else EGONN_CONV0_LAUNCH(true, false);
  } else {                                                 // unit features: exact bf16 x 3 split of the kernel
    if (out_bf16)
      hipLaunchKernelGGL((conv0_k5_unit_kernel<true>), dim3(grid), dim3(256), 0, stream, V.keys, P.g0, P.t2m, ctx->dev_counts,
                         (int32_t)P.cap[2], (int32_t)P.cap[0], W, scale, shift, relu, out, ctx->conv0_lut,
                         reinterpret_cast<const uint4*>(wpk));
    else
      hipLaunchKernelGGL((conv0_k5_unit_kernel<false>), dim3(grid), dim3(256), 0, stream, V.keys, P.g0, P.t2m, ctx->dev_counts,
                         (int32_t)P.cap[2], (int32_t)P.cap[0], W, scale, shift, relu, out, ctx->conv0_lut,
                         reinterpret_cast<const uint4*>(wpk));
  }
#undef EGONN_CONV0_LAUNCH
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

}  // namespace egonn
