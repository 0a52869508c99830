// Sparse convolution kernels for gfx950 (fp32, exact-f32 MFMA).
//
// Replaces MinkowskiConvolution / MinkowskiConvolutionTranspose forward as called from the reference
// (models/minkgl.py:39,100,105; ME BasicBlock conv1/conv2 via layers/eca_block.py:58-63).
//
//   out[o] = sum_k  in[ nbr[o][k] ] @ W[k]          (nbr[o][k] = -1: no contribution)
//
// sconv_mfma_kernel — output-stationary, pair-compacted gather -> MFMA -> LDS accumulate:
//   * a workgroup owns T consecutive output rows (Z-order => spatially clustered => the gathered
//     input rows are L2-local), keeps their fp32 accumulators in LDS and writes each output row once
//     (no HBM atomics, deterministic);
//   * for every kernel offset k the tile's valid (row, input) pairs are compacted with
//     ballot/popcount, so the MFMA M-dimension only carries real pairs: flops = 2*P*Cin*Cout (+ padding
//     to 16), not 2*27*N*Cin*Cout;
//   * 16 gathered rows are staged in LDS ([16][Cin+4], b128 reads), W[k] lives in B-fragment registers
//     of the wave that owns the column slice, v_mfma_f32_16x16x4_f32 accumulates, results are added
//     into the LDS accumulator rows of the pairs (rows are distinct within a chunk, column slices are
//     owned by one wave => plain read-modify-write, no atomics);
//   * epilogue: folded BatchNorm scale/shift (+ReLU) fused, one coalesced float4 store per element.
//
// conv0_k5_kernel — the 5x5x5, Cin=1 first layer: pure lookup work.  One wave per 4x4x4 block, the 27
// adjacent blocks' occupancy masks sit in LDS, 125 offsets are bit tests + popcounts spread over the
// lanes, hits are reduced against the 125x32 weight table in LDS.  No kernel map is materialised.
#include <algorithm>

#include "common.h"
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

// ------------------------------------------------------------------ MFMA kernel
template <int CIN, int COUT>
struct SconvCfg {
  static constexpr int NW = (COUT >= 128) ? 32 : 16;      // columns per wave
  static constexpr int NT = NW / 16;                      // 16-wide MFMA column tiles per wave
  static constexpr int WAVES_N = COUT / NW;               // waves across the columns
  static constexpr int WAVES_M = 4 / WAVES_N;             // chunk groups working concurrently
  static constexpr int CPS = (128 / CIN) / WAVES_M > 0 ? (128 / CIN) / WAVES_M : 1;   // chunks per group per step
  static constexpr int STAGE_ROWS = WAVES_M * CPS * 16;   // gathered rows per step
  static constexpr int LDA = CIN + 4;                     // A-stage row stride (floats, 16-B aligned)
  static constexpr int LDC = COUT + 4;                    // accumulator row stride
  static constexpr int KSTEPS = CIN / 16;                 // b128 A reads per chunk
  static_assert(COUT % NW == 0 && 4 % WAVES_N == 0, "bad tiling");
};

template <int CIN, int COUT>
static size_t sconv_lds_bytes(int K, int T) {
  using C = SconvCfg<CIN, COUT>;
  size_t b = 0;
  b += (size_t)C::WAVES_M * T * C::LDC * 4;     // accumulators (one copy per chunk group: no cross-wave RMW)
  b += std::max((size_t)2 * C::STAGE_ROWS * C::LDA * 4,   // A stage (double buffered) ...
                (size_t)T * K * 4);                       // ... aliased with the raw neighbour-table tile (phase A only)
  b += (size_t)T * K * 4;                       // pair input rows
  b += (size_t)T * K;                           // pair output rows (u8)
  b += (size_t)(K + 1) * 4 * 2;                 // cnt, cbase
  b += (size_t)(K * (T / 16) + 1) * 2;          // chunk table (k, c)
  return align_up(b, 16) + 64;
}

// grid = (tiles, nsplit).  Split `blockIdx.y` handles kernel offsets k with k % nsplit == blockIdx.y and, when
// nsplit > 1, writes raw partial sums to out + split * n_out * COUT (the caller reduces them in fixed order).
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void sconv_mfma_kernel(const float* __restrict__ in,
                                                          const int32_t* __restrict__ nbr,
                                                          const float* __restrict__ W,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int relu,
                                                          float* __restrict__ out, int32_t n_out, int K, int T) {
  using C = SconvCfg<CIN, COUT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* accL = reinterpret_cast<float*>(smem);                               // [WAVES_M][T][LDC]
  float* As = accL + (size_t)C::WAVES_M * T * C::LDC;                         // [2][STAGE_ROWS][LDA]
  int32_t* tbl = reinterpret_cast<int32_t*>(As);                              // [T][K], dead before As is written
  const size_t stage_words = max((size_t)2 * C::STAGE_ROWS * C::LDA, (size_t)T * K);
  int32_t* pj = reinterpret_cast<int32_t*>(As) + stage_words;                 // [K][T]
  int32_t* cnt = pj + (size_t)K * T;                                          // [K+1]
  int32_t* cbase = cnt + (K + 1);                                             // [K+1]
  uint8_t* pr = reinterpret_cast<uint8_t*>(cbase + (K + 1));                  // [K][T]
  uint8_t* ck = pr + (size_t)K * T;                                           // chunk -> k
  uint8_t* cc = ck + (size_t)K * (T / 16);                                    // chunk -> index inside k

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int32_t row0 = blockIdx.x * T;
  const int32_t rows = min(T, n_out - row0);
  const int nsplit = gridDim.y, split = blockIdx.y;

  // ---- zero accumulators, stage the tile of the neighbour table (coalesced)
  for (int i = tid; i < C::WAVES_M * T * C::LDC / 4; i += 256)
    reinterpret_cast<float4*>(accL)[i] = make_float4(0, 0, 0, 0);
  {
    const int32_t* src = nbr + (int64_t)row0 * K;
    const int n = rows * K;
    for (int i = tid; i < n; i += 256) tbl[i] = src[i];
  }
  __syncthreads();

  // ---- phase A: compact the tile's (row, input) pairs per offset with ballot/popcount
  {
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int k = wave; k < K; k += 4) {
      int32_t running = 0;
      if (k % nsplit == split) {
        for (int base = 0; base < T; base += 64) {
          const int r = base + lane;
          const int32_t j = (r < rows) ? tbl[r * K + k] : -1;
          const uint64_t m = __ballot(j >= 0);
          if (j >= 0) {
            const int pos = running + __popcll(m & lt);
            pj[k * T + pos] = j;
            pr[k * T + pos] = (uint8_t)r;
          }
          running += __popcll(m);
        }
      }
      if (lane == 0) cnt[k] = running;
    }
  }
  __syncthreads();
  if (tid == 0) {
    int32_t q = 0;
    for (int k = 0; k < K; ++k) {
      cbase[k] = q;
      const int nc = (cnt[k] + 15) >> 4;
      for (int c = 0; c < nc; ++c) {
        ck[q] = (uint8_t)k;
        cc[q] = (uint8_t)c;
        ++q;
      }
    }
    cbase[K] = q;
  }
  __syncthreads();
  const int total_chunks = cbase[K];
  constexpr int CHUNKS_PER_STEP = C::WAVES_M * C::CPS;
  const int steps = (total_chunks + CHUNKS_PER_STEP - 1) / CHUNKS_PER_STEP;

  const int grp = wave / C::WAVES_N;          // chunk group of this wave
  const int nsl = wave % C::WAVES_N;          // column slice of this wave
  const int n0 = nsl * C::NW;
  const int l15 = lane & 15, g4 = lane >> 4;

  // gather assignment: thread -> (row of the step's A stage, float4 column)
  constexpr int F4_PER_ROW = CIN / 4;
  constexpr int GATHER_ITERS = (C::STAGE_ROWS * F4_PER_ROW + 255) / 256;

  float4 greg[GATHER_ITERS];
  auto gather_issue = [&](int step) {
#pragma unroll
    for (int it = 0; it < GATHER_ITERS; ++it) {
      const int e = it * 256 + tid;
      const int srow = e / F4_PER_ROW, c4 = e - srow * F4_PER_ROW;
      float4 v = make_float4(0, 0, 0, 0);
      if (srow < C::STAGE_ROWS) {
        const int q = step * CHUNKS_PER_STEP + (srow >> 4);     // stage row block b <-> chunk q = step*CPS_total + b
        if (q < total_chunks) {
          const int k = ck[q], c = cc[q];
          const int p = c * 16 + (srow & 15);
          if (p < cnt[k]) {
            const int32_t j = pj[k * T + p];
            v = reinterpret_cast<const float4*>(in + (int64_t)j * CIN)[c4];
          }
        }
      }
      greg[it] = v;
    }
  };
  auto gather_commit = [&](int buf) {
#pragma unroll
    for (int it = 0; it < GATHER_ITERS; ++it) {
      const int e = it * 256 + tid;
      const int srow = e / F4_PER_ROW, c4 = e - srow * F4_PER_ROW;
      if (srow < C::STAGE_ROWS)
        *reinterpret_cast<float4*>(As + ((size_t)buf * C::STAGE_ROWS + srow) * C::LDA + c4 * 4) = greg[it];
    }
  };

  float breg[C::NT][CIN / 4];
  int cur_k = -1;
  float* myacc = accL + (size_t)grp * T * C::LDC;

  if (steps > 0) {
    gather_issue(0);
    gather_commit(0);
  }
  __syncthreads();

  for (int s = 0; s < steps; ++s) {
    const int buf = s & 1;
    if (s + 1 < steps) gather_issue(s + 1);
#pragma unroll
    for (int ci = 0; ci < C::CPS; ++ci) {
      const int blk = grp * C::CPS + ci;                 // block of 16 stage rows == chunk inside the step
      const int q = s * CHUNKS_PER_STEP + blk;
      if (q < total_chunks) {
        const int k = ck[q], c = cc[q];
        if (k != cur_k) {
          cur_k = k;
          const float* wk = W + (size_t)k * CIN * COUT;
#pragma unroll
          for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
            for (int t = 0; t < C::KSTEPS; ++t)
#pragma unroll
              for (int u = 0; u < 4; ++u)
                breg[nt][t * 4 + u] = wk[(size_t)(16 * t + 4 * g4 + u) * COUT + n0 + nt * 16 + l15];
        }
        f32x4 acc[C::NT];
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* arow = As + ((size_t)buf * C::STAGE_ROWS + blk * 16 + l15) * C::LDA + 4 * g4;
#pragma unroll
        for (int t = 0; t < C::KSTEPS; ++t) {
          const float4 a4 = *reinterpret_cast<const float4*>(arow + 16 * t);
          const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt)
              acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], breg[nt][t * 4 + u], acc[nt], 0, 0, 0);
        }
        // accumulate: D[row = 4*g4 + r][col = l15] -> LDS accumulator row of the pair (rows are distinct
        // inside a chunk, the column slice belongs to this wave, the copy to this group: no conflicts)
        const int nvalid = cnt[k] - c * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int p = 4 * g4 + r;
          if (p < nvalid) {
            const int orow = pr[k * T + c * 16 + p];
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt) myacc[orow * C::LDC + n0 + nt * 16 + l15] += acc[nt][r];
          }
        }
      }
    }
    if (s + 1 < steps) gather_commit(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: (split: raw partial sums) | BN scale/shift (+ReLU); one coalesced store per element
  constexpr int O4 = COUT / 4;
  float* dst = out + (nsplit > 1 ? (size_t)split * n_out * COUT : 0);
  for (int e = tid; e < rows * O4; e += 256) {
    const int r = e / O4, c4 = e - r * O4;
    float4 v = *reinterpret_cast<const float4*>(accL + r * C::LDC + c4 * 4);
#pragma unroll
    for (int g = 1; g < C::WAVES_M; ++g) {
      const float4 w = *reinterpret_cast<const float4*>(accL + ((size_t)g * T + r) * C::LDC + c4 * 4);
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    if (nsplit == 1) {
      if (scale) {
        const float4 sc = reinterpret_cast<const float4*>(scale)[c4];
        const float4 sh = reinterpret_cast<const float4*>(shift)[c4];
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
      }
      if (relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
    }
    reinterpret_cast<float4*>(dst + (int64_t)(row0 + r) * COUT)[c4] = v;
  }
}

// out[e] = act( (sum_s partial[s][e]) * scale + shift ), fixed summation order => deterministic
__global__ void sconv_reduce_kernel(const float* __restrict__ partial, int nsplit, int64_t n4, int c4n,
                                    const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                    float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n4) return;
  float4 v = reinterpret_cast<const float4*>(partial)[e];
  for (int s = 1; s < nsplit; ++s) {
    const float4 w = reinterpret_cast<const float4*>(partial)[(int64_t)s * n4 + e];
    v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
  }
  if (scale) {
    const int c4 = (int)(e % c4n);
    const float4 sc = reinterpret_cast<const float4*>(scale)[c4];
    const float4 sh = reinterpret_cast<const float4*>(shift)[c4];
    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
  }
  if (relu) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  }
  reinterpret_cast<float4*>(out)[e] = v;
}

template <int CIN, int COUT>
static int launch_sconv(const float* in, const int32_t* nbr, const float* W, const float* scale, const float* shift,
                        int relu, float* out, int32_t n_out, int K, float* scratch, size_t scratch_floats,
                        hipStream_t stream) {
  // tile: 128 rows when there is enough work to fill the chip twice over, otherwise 64
  const int T = (n_out >= 128 * 512) ? 128 : 64;
  const int tiles = (int)cdiv(n_out, T);
  // small levels are latency bound: split the kernel offsets over workgroups until the chip is busy
  int nsplit = 1;
  if (scratch && tiles < 384) {
    nsplit = (int)std::min<int64_t>(K, cdiv(512, tiles));
    while (nsplit > 1 && (size_t)nsplit * n_out * COUT > scratch_floats) --nsplit;
  }
  const size_t lds = sconv_lds_bytes<CIN, COUT>(K, T);
  static bool attr_done = false;
  if (!attr_done) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sconv_mfma_kernel<CIN, COUT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  float* dst = nsplit > 1 ? scratch : out;
  hipLaunchKernelGGL((sconv_mfma_kernel<CIN, COUT>), dim3((unsigned)tiles, (unsigned)nsplit), dim3(256), lds, stream,
                     in, nbr, W, scale, shift, relu, dst, n_out, K, T);
  if (nsplit > 1) {
    const int64_t n4 = (int64_t)n_out * COUT / 4;
    hipLaunchKernelGGL(sconv_reduce_kernel, dim3((unsigned)cdiv(n4, 256)), dim3(256), 0, stream, scratch, nsplit, n4,
                       COUT / 4, scale, shift, relu, out);
  }
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

static bool g_force_naive = false;
void sconv_set_naive(bool on) { g_force_naive = on; }

int sconv_forward(const float* in, const int32_t* nbr, const float* W, const float* scale, const float* shift,
                  int relu, float* out, int32_t n_out, int K, int cin, int cout, float* scratch,
                  size_t scratch_floats, hipStream_t stream) {
  if (n_out == 0) return EGONN_OK;
  EGONN_REQUIRE(K == 27 || K == 8, EGONN_ERR_INVALID, "sconv: kernel volume %d not supported", K);
  if (!g_force_naive) {
#define EGONN_SCONV_CASE(CI, CO)  \
  if (cin == CI && cout == CO)    \
    return launch_sconv<CI, CO>(in, nbr, W, scale, shift, relu, out, n_out, K, scratch, scratch_floats, stream);
    EGONN_SCONV_CASE(32, 32)
    EGONN_SCONV_CASE(32, 64)
    EGONN_SCONV_CASE(64, 64)
    EGONN_SCONV_CASE(64, 128)
    EGONN_SCONV_CASE(128, 128)
#undef EGONN_SCONV_CASE
  }
  const int64_t total = (int64_t)n_out * cout;
  hipLaunchKernelGGL(sconv_naive_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, in, nbr, W, scale,
                     shift, relu, out, n_out, K, cin, cout);
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

__global__ __launch_bounds__(256) void conv0_k5_kernel(const float* __restrict__ feat,         // [n0] (Cin = 1)
                                                        const uint64_t* __restrict__ vkeys,     // level 0
                                                        const int32_t* __restrict__ parent0,    // level 0 -> 1
                                                        const int32_t* __restrict__ parent1,    // level 1 -> 2
                                                        const int32_t* __restrict__ badj,       // [n2][27]
                                                        const uint64_t* __restrict__ bmask,
                                                        const int32_t* __restrict__ bstart, int32_t nvox,
                                                        int32_t ntiles, const float* __restrict__ W,   // [125][32]
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int relu,
                                                        float* __restrict__ out,
                                                        unsigned long long* __restrict__ pair_counter) {
  __shared__ uint64_t s_m[4][16][27];
  __shared__ int32_t s_s[4][16][27];
  __shared__ uint32_t s_koff[128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g4 = lane >> 4;
  if (tid < 128) {
    const int k = tid;
    s_koff[k] = (k < 125) ? (uint32_t)((k % 5) | (((k / 5) % 5) << 4) | ((k / 25) << 8)) : 0xFFFFu;
  }
  float breg[2][32];
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int k = 16 * (q >> 2) + 4 * g4 + (q & 3);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) breg[nt][q] = (k < 125) ? W[k * COUT0 + nt * 16 + l15] : 0.f;
  }
  float sc[2], sh[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    sc[nt] = scale ? scale[nt * 16 + l15] : 1.f;
    sh[nt] = scale ? shift[nt * 16 + l15] : 0.f;
  }
  __syncthreads();
  int32_t npairs = 0;
  for (int32_t tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
    const int32_t r0 = tile * 16;
    // ---- (mask, first row) of the 27 blocks around every row's block
    for (int e = lane; e < 16 * 27; e += 64) {
      const int row = e / 27, slot = e - row * 27;
      const int32_t r = r0 + row;
      uint64_t m = 0;
      int32_t st = 0;
      if (r < nvox) {
        const int32_t b = parent1[parent0[r]];
        const int32_t adj = badj[(int64_t)b * 27 + slot];
        if (adj >= 0) {
          m = bmask[adj];
          st = bstart[adj];
        }
      }
      s_m[wave][row][slot] = m;
      s_s[wave][row][slot] = st;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int32_t r = r0 + l15;
    const bool vrow = r < nvox;
    const uint32_t lk = vrow ? (uint32_t)(vkeys[r] & 63) : 0u;
    const int32_t lx = (lk & 1) | ((lk >> 2) & 2), ly = ((lk >> 1) & 1) | ((lk >> 3) & 2),
                  lz = ((lk >> 2) & 1) | ((lk >> 4) & 2);
    float a[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const int k = 16 * (q >> 2) + 4 * g4 + (q & 3);
      const uint32_t po = s_koff[k];
      float v = 0.f;
      if (vrow && po != 0xFFFFu) {
        const int32_t nx = lx + (int32_t)(po & 15) - 2, ny = ly + (int32_t)((po >> 4) & 15) - 2,
                      nz = lz + (int32_t)(po >> 8) - 2;
        const int32_t slot = ((nx < 0) ? 0 : (nx > 3 ? 2 : 1)) + 3 * ((ny < 0) ? 0 : (ny > 3 ? 2 : 1)) +
                             9 * ((nz < 0) ? 0 : (nz > 3 ? 2 : 1));
        const uint32_t ux = (uint32_t)nx & 3, uy = (uint32_t)ny & 3, uz = (uint32_t)nz & 3;
        const uint32_t bit = (ux & 1) | ((uy & 1) << 1) | ((uz & 1) << 2) | ((ux & 2) << 2) | ((uy & 2) << 3) |
                             ((uz & 2) << 4);
        const uint64_t m = s_m[wave][l15][slot];
        if ((m >> bit) & 1) {
          v = feat[s_s[wave][l15][slot] + __popcll(m & ((1ull << bit) - 1))];
          ++npairs;
        }
      }
      a[q] = v;
    }
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], breg[0][q], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], breg[1][q], acc[1], 0, 0, 0);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int32_t orow = r0 + 4 * g4 + rr;
      if (orow < nvox) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          float o = acc[nt][rr] * sc[nt] + sh[nt];
          if (relu) o = fmaxf(o, 0.f);
          out[(int64_t)orow * COUT0 + nt * 16 + l15] = o;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) npairs += __shfl_xor(npairs, o, 64);
  if (lane == 0 && npairs) atomicAdd(pair_counter, (unsigned long long)npairs);
}

int conv0_k5_forward(Ctx* ctx, const float* feat, const float* W, int cout, const float* scale,
                     const float* shift, int relu, float* out, hipStream_t stream) {
  const Plan& P = ctx->plan;
  EGONN_REQUIRE(cout == COUT0, EGONN_ERR_INVALID, "conv0: %d output channels not supported (expected %d)", cout, COUT0);
  const Level& V = P.lv[0];
  const Level& B = P.lv[2];
  if (V.n == 0) return EGONN_OK;
  const int32_t ntiles = (int32_t)cdiv(V.n, 16);
  const unsigned grid = (unsigned)std::min<int64_t>(cdiv(ntiles, 4), 1536);
  hipLaunchKernelGGL(conv0_k5_kernel, dim3(grid), dim3(256), 0, stream, feat, V.keys, V.parent, P.lv[1].parent,
                     B.nbr27, B.mask, B.bstart, (int32_t)V.n, ntiles, W, scale, shift, relu, out, ctx->dev_pairs);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

}  // namespace egonn
