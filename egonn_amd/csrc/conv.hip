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
  static constexpr int WAVES_M = 4 / WAVES_N;             // chunks processed per step
  static constexpr int LDA = CIN + 4;                     // A-stage row stride (floats, 16-B aligned)
  static constexpr int LDC = COUT + 4;                    // accumulator row stride
  static constexpr int KSTEPS = CIN / 16;                 // b128 A reads per chunk
  static_assert(COUT % NW == 0 && 4 % WAVES_N == 0, "bad tiling");
};

static size_t sconv_lds_bytes(int cin, int cout, int K, int T) {
  const int wn = cout / (cout >= 128 ? 32 : 16);
  const int wm = 4 / wn;
  size_t b = 0;
  b += (size_t)wm * T * (cout + 4) * 4;         // acc (one copy per chunk group)
  b += (size_t)2 * wm * 16 * (cin + 4) * 4;     // A stage (double buffered)
  b += (size_t)T * K * 4;                       // tile of the neighbour table / pair input rows
  b += (size_t)T * K;                           // pair output rows (u8)
  b += (size_t)(K + 1) * 4 * 2;                 // cnt, cbase
  b += (size_t)(K * (T / 16) + 1) * 2;          // chunk table (k, c)
  return align_up(b, 16) + 64;
}

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
  float* As = accL + (size_t)C::WAVES_M * T * C::LDC;                         // [2][WAVES_M*16][LDA]
  int32_t* pj = reinterpret_cast<int32_t*>(As + 2 * C::WAVES_M * 16 * C::LDA);   // [K][T] (first: raw table [T][K])
  int32_t* cnt = pj + (size_t)K * T;                                          // [K+1]
  int32_t* cbase = cnt + (K + 1);                                             // [K+1]
  uint8_t* pr = reinterpret_cast<uint8_t*>(cbase + (K + 1));                  // [K][T]
  uint8_t* ck = pr + (size_t)K * T;                                           // [K*T/16] chunk -> k
  uint8_t* cc = ck + (size_t)K * (T / 16);                                    // chunk -> index inside k

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int32_t row0 = blockIdx.x * T;
  const int32_t rows = min(T, n_out - row0);

  // ---- zero accumulators
  for (int i = tid; i < C::WAVES_M * T * C::LDC / 4; i += 256)
    reinterpret_cast<float4*>(accL)[i] = make_float4(0, 0, 0, 0);

  // ---- phase A: compact the tile's (row, input) pairs per offset.  The raw [rows][K] table is read
  //      straight from global (each wave reads the k-column of 64 rows: strided, L2-resident, 4*K B/row).
  {
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int k = wave; k < K; k += 4) {
      int32_t running = 0;
      for (int base = 0; base < T; base += 64) {
        const int r = base + lane;
        const int32_t j = (r < rows) ? nbr[(int64_t)(row0 + r) * K + k] : -1;
        const uint64_t m = __ballot(j >= 0);
        if (j >= 0) {
          const int pos = running + __popcll(m & lt);
          pj[k * T + pos] = j;
          pr[k * T + pos] = (uint8_t)r;
        }
        running += __popcll(m);
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
  const int steps = (total_chunks + C::WAVES_M - 1) / C::WAVES_M;

  const int grp = wave / C::WAVES_N;          // which chunk of the step this wave works on
  const int nsl = wave % C::WAVES_N;          // which column slice
  const int n0 = nsl * C::NW;
  const int l15 = lane & 15, g4 = lane >> 4;

  // gather assignment: thread -> (row of the step's A stage, float4 column)
  constexpr int F4_PER_ROW = CIN / 4;
  constexpr int STAGE_ROWS = C::WAVES_M * 16;
  constexpr int GATHER_ITERS = (STAGE_ROWS * F4_PER_ROW + 255) / 256;

  float4 greg[GATHER_ITERS];
  auto gather_issue = [&](int step) {
#pragma unroll
    for (int it = 0; it < GATHER_ITERS; ++it) {
      const int e = it * 256 + tid;
      const int srow = e / F4_PER_ROW, c4 = e - srow * F4_PER_ROW;
      float4 v = make_float4(0, 0, 0, 0);
      if (srow < STAGE_ROWS) {
        const int q = step * C::WAVES_M + (srow >> 4);
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
      if (srow < STAGE_ROWS)
        *reinterpret_cast<float4*>(As + ((size_t)buf * STAGE_ROWS + srow) * C::LDA + c4 * 4) = greg[it];
    }
  };

  float breg[C::NT][CIN / 4];
  int cur_k = -1;

  if (steps > 0) {
    gather_issue(0);
    gather_commit(0);
  }
  __syncthreads();

  for (int s = 0; s < steps; ++s) {
    const int buf = s & 1;
    if (s + 1 < steps) gather_issue(s + 1);

    const int q = s * C::WAVES_M + grp;
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
      const float* arow = As + ((size_t)buf * STAGE_ROWS + grp * 16 + l15) * C::LDA + 4 * g4;
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
      // scatter-add: D[row = 4*g4 + r][col = l15]
      const int nvalid = cnt[k] - c * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = 4 * g4 + r;
        if (p < nvalid) {
          const int orow = pr[k * T + c * 16 + p];
#pragma unroll
          for (int nt = 0; nt < C::NT; ++nt)
            accL[((size_t)grp * T + orow) * C::LDC + n0 + nt * 16 + l15] += acc[nt][r];
        }
      }
    }
    if (s + 1 < steps) gather_commit(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: BN scale/shift (+ReLU), one coalesced store per element
  constexpr int O4 = COUT / 4;
  for (int e = tid; e < rows * O4; e += 256) {
    const int r = e / O4, c4 = e - r * O4;
    float4 v = *reinterpret_cast<const float4*>(accL + r * C::LDC + c4 * 4);
#pragma unroll
    for (int g = 1; g < C::WAVES_M; ++g) {
      const float4 w = *reinterpret_cast<const float4*>(accL + ((size_t)g * T + r) * C::LDC + c4 * 4);
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    if (scale) {
      const float4 sc = reinterpret_cast<const float4*>(scale)[c4];
      const float4 sh = reinterpret_cast<const float4*>(shift)[c4];
      v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
    }
    if (relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    reinterpret_cast<float4*>(out + (int64_t)(row0 + r) * COUT)[c4] = v;
  }
}

template <int CIN, int COUT>
static int launch_sconv(const float* in, const int32_t* nbr, const float* W, const float* scale, const float* shift,
                        int relu, float* out, int32_t n_out, int K, hipStream_t stream) {
  // tile: 128 rows when there is enough work to fill the chip twice over, otherwise 64
  const int T = (n_out >= 128 * 512) ? 128 : 64;
  const size_t lds = sconv_lds_bytes(CIN, COUT, K, T);
  static bool attr_done = false;
  if (!attr_done) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sconv_mfma_kernel<CIN, COUT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL((sconv_mfma_kernel<CIN, COUT>), dim3((unsigned)cdiv(n_out, T)), dim3(256), lds, stream, in, nbr,
                     W, scale, shift, relu, out, n_out, K, T);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

static bool g_force_naive = false;
void sconv_set_naive(bool on) { g_force_naive = on; }

int sconv_forward(const float* in, const int32_t* nbr, const float* W, const float* scale, const float* shift,
                  int relu, float* out, int32_t n_out, int K, int cin, int cout, hipStream_t stream) {
  if (n_out == 0) return EGONN_OK;
  EGONN_REQUIRE(K == 27 || K == 8, EGONN_ERR_INVALID, "sconv: kernel volume %d not supported", K);
  if (!g_force_naive) {
#define EGONN_SCONV_CASE(CI, CO) \
  if (cin == CI && cout == CO) return launch_sconv<CI, CO>(in, nbr, W, scale, shift, relu, out, n_out, K, stream);
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
static constexpr int COUT0 = 32;

__device__ static inline uint32_t bit_of_local5(uint32_t lx, uint32_t ly, uint32_t lz) {
  return (lx & 1) | ((ly & 1) << 1) | ((lz & 1) << 2) | ((lx & 2) << 2) | ((ly & 2) << 3) | ((lz & 2) << 4);
}

__device__ static inline int32_t find_key5(const uint64_t* __restrict__ keys, int32_t n, uint64_t q) {
  int32_t lo = 0, hi = n;
  while (lo < hi) {
    const int32_t mid = (lo + hi) >> 1;
    if (keys[mid] < q) lo = mid + 1; else hi = mid;
  }
  return (lo < n && keys[lo] == q) ? lo : -1;
}

__global__ __launch_bounds__(256) void conv0_k5_kernel(const float* __restrict__ feat,        // [n0] (Cin = 1)
                                                        const uint64_t* __restrict__ vkeys,    // level 0
                                                        const uint64_t* __restrict__ bkeys,    // level 2
                                                        const uint64_t* __restrict__ bmask,
                                                        const int32_t* __restrict__ bstart, int32_t nblocks,
                                                        int32_t nvox, int cbL, const float* __restrict__ W,   // [125][32]
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int relu,
                                                        float* __restrict__ out,
                                                        unsigned long long* __restrict__ pair_counter) {
  __shared__ float sW[125 * COUT0];
  __shared__ uint64_t s_m[4][27];
  __shared__ int32_t s_s[4][27];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 125 * COUT0; i += 256) sW[i] = W[i];
  const int32_t j = blockIdx.x * 4 + wave;
  if (j < nblocks && lane < 27) {
    const uint64_t key = bkeys[j];
    const uint64_t mort = key & ((1ull << (3 * cbL)) - 1);
    const uint64_t bat = key >> (3 * cbL);
    const int32_t bx = (int32_t)compact1by2(mort), by = (int32_t)compact1by2(mort >> 1),
                  bz = (int32_t)compact1by2(mort >> 2);
    const int32_t nx = bx + (lane % 3) - 1, ny = by + (lane / 3) % 3 - 1, nz = bz + lane / 9 - 1;
    const int32_t lim = 1 << cbL;
    int32_t idx = -1;
    if (lane == 13) idx = j;
    else if (nx >= 0 && nx < lim && ny >= 0 && ny < lim && nz >= 0 && nz < lim)
      idx = find_key5(bkeys, nblocks, (bat << (3 * cbL)) | morton3((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
    s_m[wave][lane] = idx >= 0 ? bmask[idx] : 0ull;
    s_s[wave][lane] = idx >= 0 ? bstart[idx] : 0;
  }
  __syncthreads();
  if (j >= nblocks) return;
  const int32_t s = bstart[j];
  const int32_t e = (j + 1 < nblocks) ? bstart[j + 1] : nvox;
  const int c = lane & 31, half = lane >> 5;
  const float sc = scale ? scale[c] : 1.f, sh = scale ? shift[c] : 0.f;
  // the two offsets this lane tests for every voxel
  const int k0 = lane, k1 = lane + 64;
  const int dx0 = k0 % 5 - 2, dy0 = (k0 / 5) % 5 - 2, dz0 = k0 / 25 - 2;
  const int dx1 = k1 % 5 - 2, dy1 = (k1 / 5) % 5 - 2, dz1 = k1 / 25 - 2;
  int32_t npairs = 0;
  for (int32_t v = s; v < e; ++v) {
    const uint32_t lk = (uint32_t)(vkeys[v] & 63);
    const int32_t lx = (lk & 1) | ((lk >> 2) & 2), ly = ((lk >> 1) & 1) | ((lk >> 3) & 2),
                  lz = ((lk >> 2) & 1) | ((lk >> 4) & 2);
    float f0 = 0.f, f1 = 0.f;
    bool h0 = false, h1 = false;
    {
      const int32_t nx = lx + dx0, ny = ly + dy0, nz = lz + dz0;
      const int32_t slot = ((nx < 0) ? 0 : (nx > 3 ? 2 : 1)) + 3 * ((ny < 0) ? 0 : (ny > 3 ? 2 : 1)) +
                           9 * ((nz < 0) ? 0 : (nz > 3 ? 2 : 1));
      const uint32_t bit = bit_of_local5((uint32_t)nx & 3, (uint32_t)ny & 3, (uint32_t)nz & 3);
      const uint64_t m = s_m[wave][slot];
      if ((m >> bit) & 1) {
        h0 = true;
        f0 = feat[s_s[wave][slot] + __popcll(m & ((1ull << bit) - 1))];
      }
    }
    if (k1 < 125) {
      const int32_t nx = lx + dx1, ny = ly + dy1, nz = lz + dz1;
      const int32_t slot = ((nx < 0) ? 0 : (nx > 3 ? 2 : 1)) + 3 * ((ny < 0) ? 0 : (ny > 3 ? 2 : 1)) +
                           9 * ((nz < 0) ? 0 : (nz > 3 ? 2 : 1));
      const uint32_t bit = bit_of_local5((uint32_t)nx & 3, (uint32_t)ny & 3, (uint32_t)nz & 3);
      const uint64_t m = s_m[wave][slot];
      if ((m >> bit) & 1) {
        h1 = true;
        f1 = feat[s_s[wave][slot] + __popcll(m & ((1ull << bit) - 1))];
      }
    }
    uint64_t m0 = __ballot(h0), m1 = __ballot(h1);
    npairs += __popcll(m0) + __popcll(m1);
    float acc = 0.f;
    // two hits per iteration: lanes 0-31 take the first, lanes 32-63 the second
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      uint64_t m = pass ? m1 : m0;
      const float fsrc = pass ? f1 : f0;
      const int kofs = pass ? 64 : 0;
      while (m) {
        const int la = __builtin_ctzll(m);
        m &= m - 1;
        int lb = -1;
        if (m) {
          lb = __builtin_ctzll(m);
          m &= m - 1;
        }
        const float fa = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fsrc), la));
        const float fb = (lb >= 0)
                             ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, fsrc), lb))
                             : 0.f;
        const int kk = half ? lb : la;
        const float ff = half ? fb : fa;
        if (kk >= 0) acc = fmaf(ff, sW[(kofs + kk) * COUT0 + c], acc);
      }
    }
    acc += __shfl_xor(acc, 32, 64);
    if (half == 0) {
      float r = acc * sc + sh;
      if (relu) r = fmaxf(r, 0.f);
      out[(int64_t)v * COUT0 + c] = r;
    }
  }
  if (lane == 0 && npairs) atomicAdd(pair_counter, (unsigned long long)npairs);
}

int conv0_k5_forward(Ctx* ctx, const float* feat, const float* W, int cout, const float* scale,
                     const float* shift, int relu, float* out, hipStream_t stream) {
  const Plan& P = ctx->plan;
  EGONN_REQUIRE(cout == COUT0, EGONN_ERR_INVALID, "conv0: %d output channels not supported (expected %d)", cout, COUT0);
  const Level& V = P.lv[0];
  const Level& B = P.lv[2];
  if (V.n == 0) return EGONN_OK;
  hipLaunchKernelGGL(conv0_k5_kernel, dim3((unsigned)cdiv(B.n, 4)), dim3(256), 0, stream, feat, V.keys, B.keys, B.mask,
                     B.bstart, (int32_t)B.n, (int32_t)V.n, P.coord_bits - 2, W, scale, shift, relu, out, ctx->dev_pairs);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

}  // namespace egonn
