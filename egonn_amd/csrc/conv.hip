// Sparse convolution kernels for gfx950 (fp32, exact-f32 MFMA).
//
// Replaces MinkowskiConvolution / MinkowskiConvolutionTranspose forward as called from the reference
// (models/minkgl.py:39,100,105; ME BasicBlock conv1/conv2 via layers/eca_block.py:58-63).
//
//   out[o] = sum_k  in[ nbr[o][k] ] @ W[k]          (nbr[o][k] = -1: no contribution)
//
// sconv_mfma_kernel — output-stationary, pair-compacted, barrier-free gather -> MFMA -> LDS accumulate:
//   * a workgroup owns T consecutive output rows (Z-order => spatially clustered => the gathered
//     input rows are L2-local), keeps their fp32 accumulators in LDS and writes each output row once
//     (no HBM atomics, deterministic);
//   * for every kernel offset k the tile's valid (row, input) pairs are compacted with
//     ballot/popcount, so the MFMA M-dimension only carries real pairs: flops = 2*P*Cin*Cout (+ padding
//     to 16), not 2*27*N*Cin*Cout;
//   * the 16 gathered rows of a chunk go straight from global memory into the MFMA A-operand registers
//     (float4 per lane, prefetched two chunks ahead), W[k] lives in B-fragment registers of the wave that
//     owns the column slice, v_mfma_f32_16x16x4_f32 accumulates, results are added into the LDS
//     accumulator rows of the pairs (rows are distinct within a chunk, column slices are owned by one
//     wave, chunk groups own separate copies => plain read-modify-write, no atomics, no barriers);
//   * epilogue: folded BatchNorm scale/shift (+ReLU) fused, one coalesced float4 store per element.
//
// conv0_k5_kernel — the 5x5x5, Cin=1 first layer: pure lookup work.  One wave per 4x4x4 block, the 27
// adjacent blocks' occupancy masks sit in LDS, 125 offsets are bit tests + popcounts spread over the
// lanes, hits are reduced against the 125x32 weight table in LDS.  No kernel map is materialised.
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

// ------------------------------------------------------------------ MFMA kernel
template <int CIN, int COUT>
struct SconvCfg {
  static constexpr int NW = (COUT <= 32) ? 16 : 32;       // columns per wave (measured: tools/bench_sconv.py)
  static constexpr int NT = NW / 16;                      // 16-wide MFMA column tiles per wave
  static constexpr int WAVES_N = COUT / NW;               // waves across the columns
  static constexpr int WAVES_M = 4 / WAVES_N;             // chunk groups working concurrently
  static constexpr int LDC = COUT + 4;                    // accumulator row stride
  static constexpr int KSTEPS = CIN / 16;                 // float4 A loads per lane per chunk
  static_assert(COUT % NW == 0 && 4 % WAVES_N == 0, "bad tiling");
};

template <int CIN, int COUT>
static size_t sconv_lds_bytes(int K, int T) {
  using C = SconvCfg<CIN, COUT>;
  size_t b = 0;
  // accumulators (+1 dummy row) per chunk group; the raw neighbour-table tile of phase A lives in the same bytes
  b += std::max((size_t)C::WAVES_M * (T + 1) * C::LDC * 4, (size_t)T * K * 4);
  b += (size_t)(T * K + 16 * K + 16) * 4;                 // pair input rows, chunk-major (+ padding)
  b += (size_t)(T * K + 16 * K + 16);                     // pair output rows (u8)
  b += (size_t)(K + 1) * 4 * 2 + 64;                      // cnt, cbase
  b += (size_t)(K * (T / 16) + K + 64);                   // chunk -> k
  return align_up(b, 16) + 64;
}

// Barrier-free gather -> MFMA -> accumulate.  grid = (tiles, nsplit): split `blockIdx.y` handles the kernel
// offsets k with k % nsplit == blockIdx.y and, when nsplit > 1, writes raw partial sums to
// out + split * n_out * COUT (sconv_reduce_kernel adds them in fixed order and applies the epilogue).
//   phase A (once per tile): the [T][K] slice of the neighbour table is staged in LDS (coalesced) and compacted
//     per offset with ballot/popcount into (input row, output row) lists padded to multiples of 16 pairs;
//   main loop (no workgroup barrier): chunks of 16 pairs are dealt round-robin to the WAVES_M wave groups; the
//     A fragment of a chunk is loaded straight from global memory into the MFMA operand registers of the lanes
//     that need it (lane (i = l & 15, g = l >> 4) loads the float4s of row pj[i] at columns 16t + 4g), two
//     chunks ahead through a register ring; W[k] fragments are reloaded when the offset changes; the 16x16
//     results are added into the group's LDS accumulator rows (rows are distinct inside a chunk, column slices
//     belong to one wave, every group has its own copy => plain read-modify-write, race-free, fixed order);
//   epilogue: group copies are summed, folded BatchNorm (+ReLU) applied, one coalesced float4 store per element.
// History (profiles/r01*): an LDS-staged variant with a barrier per step spent its time serialised (ablation:
// gather / W / accumulate / barrier / MFMA each 10-17 %); ds_add_f32 LDS atomics were 3.3x slower than the RMW.
// BF16 = true: the MFMA operands are rounded to bf16 (A fragments converted in registers after the fp32 gather, W
// pre-packed as bf16 in the same fragment order), products accumulate in fp32 on v_mfma_f32_16x16x32_bf16 — BASELINE
// configs[2]; feature maps stay fp32 in HBM.  One instruction covers two of the 16-wide K slabs of the fp32 path: the
// K index is a reduction index, so lane g simply owns columns {32s+4g..+3} and {32s+16+4g..+3} of both operands.
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
__device__ static inline uint32_t pack_bf16x2(float a, float b) {       // round to nearest even, a in the low half
  uint32_t ua = __float_as_uint(a), ub = __float_as_uint(b);
  ua += 0x7FFFu + ((ua >> 16) & 1u);
  ub += 0x7FFFu + ((ub >> 16) & 1u);
  return (ua >> 16) | (ub & 0xFFFF0000u);
}
template <int CIN, int COUT, bool BF16>
__global__ __launch_bounds__(256) void sconv_mfma_kernel(const float* __restrict__ in,
                                                          const int32_t* __restrict__ nbr,
                                                          const float* __restrict__ W,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int relu,
                                                          float* __restrict__ out, int32_t n_out, int K, int T,
                                                          uint32_t in_bytes, uint32_t w_bytes) {
  using C = SconvCfg<CIN, COUT>;
  // gather prefetch depth (chunks in flight per wave): the A rows come from L2/HBM at random-access latency
  constexpr int SLOT_F4 = BF16 ? (C::KSTEPS * (2 + C::NT) + 1) / 2 : C::KSTEPS * (1 + C::NT);   // float4 regs per ring slot (A + W)
  // (deeper rings were measured: 6 / 8 slots for the <32,32> plan run 4 % / 15 % slower — registers cost more waves
  //  than the extra lookahead hides)
  constexpr int DDEPTH = (SLOT_F4 <= 4) ? 4 : ((SLOT_F4 <= 12) ? 3 : 2);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* accL = reinterpret_cast<float*>(smem);                               // [WAVES_M][T+1][LDC]
  int32_t* tbl = reinterpret_cast<int32_t*>(accL);        // [T][K], phase A only: aliased with the accumulators
  const int list_cap = T * K + 16 * K + 16;
  const size_t acc_bytes = std::max((size_t)C::WAVES_M * (T + 1) * C::LDC * 4, (size_t)T * K * 4);
  int32_t* pj = reinterpret_cast<int32_t*>(smem + acc_bytes);                 // [chunk][16] input row or -1
  int32_t* cnt = pj + list_cap;                                               // [K+1]
  int32_t* cbase = cnt + (K + 1);                                             // [K+1] first chunk of offset k
  uint8_t* pr = reinterpret_cast<uint8_t*>(cbase + (K + 1) + 8);              // [chunk][16] output row (T = dummy)
  uint8_t* ck = pr + list_cap;                                                // chunk -> k

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int32_t row0 = blockIdx.x * T;
  const int32_t rows = min(T, n_out - row0);
  const int nsplit = gridDim.y, split = blockIdx.y;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

  // ---- stage the tile of the neighbour table (coalesced); the accumulators are zeroed after phase A
  {
    const int32_t* src = nbr + (int64_t)row0 * K;
    const int n = rows * K;
    for (int i = tid; i < n; i += 256) tbl[i] = src[i];
  }
  __syncthreads();

  // ---- phase A1: pairs per offset (ballot + popcount)
  for (int k = wave; k < K; k += 4) {
    int32_t running = 0;
    if (k % nsplit == split) {
      for (int base = 0; base < T; base += 64) {
        const int r = base + lane;
        const int32_t j = (r < rows) ? tbl[r * K + k] : -1;
        running += __popcll(__ballot(j >= 0));
      }
    }
    if (lane == 0) cnt[k] = running;
  }
  __syncthreads();
  if (tid < 64) {                      // chunk table by one wave: prefix over the <= 27 offsets with shuffles
    const int nc = (tid < K) ? ((cnt[tid] + 15) >> 4) : 0;
    int incl = nc;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    const int first = incl - nc;
    if (tid < K) {
      cbase[tid] = first;
      for (int c = 0; c < nc; ++c) ck[first + c] = (uint8_t)tid;
    }
    if (tid == K - 1) cbase[K] = incl;
  }
  __syncthreads();
  const int total_chunks = cbase[K];
  // ---- phase A2: chunk-major compacted (input row, output row) lists; chunk tails padded with (-1, dummy row T)
  for (int k = wave; k < K; k += 4) {
    if (k % nsplit != split) continue;
    const int base_e = cbase[k] * 16;
    int32_t running = 0;
    for (int base = 0; base < T; base += 64) {
      const int r = base + lane;
      const int32_t j = (r < rows) ? tbl[r * K + k] : -1;
      const uint64_t m = __ballot(j >= 0);
      if (j >= 0) {
        const int pos = base_e + running + __popcll(m & lt);
        pj[pos] = j;
        pr[pos] = (uint8_t)r;
      }
      running += __popcll(m);
    }
    const int padded = ((running + 15) >> 4) << 4;
    if (lane < padded - running) {
      pj[base_e + running + lane] = -1;
      pr[base_e + running + lane] = (uint8_t)T;
    }
  }
  if (tid < 16) {                                              // one all-padding chunk behind the last real one
    pj[total_chunks * 16 + tid] = -1;
    pr[total_chunks * 16 + tid] = (uint8_t)T;
    if (tid == 0) ck[total_chunks] = (uint8_t)(total_chunks ? ck[total_chunks - 1] : 0);
  }
  __syncthreads();
  for (int i = tid; i < C::WAVES_M * (T + 1) * C::LDC / 4; i += 256)          // the table tile is dead now
    reinterpret_cast<float4*>(accL)[i] = make_float4(0, 0, 0, 0);
  __syncthreads();

  const int grp = wave / C::WAVES_N;          // chunk group of this wave (chunks are dealt round-robin to groups)
  const int nsl = wave % C::WAVES_N;          // column slice of this wave
  const int n0 = nsl * C::NW;
  const int l15 = lane & 15, g4 = lane >> 4;
  const int my_chunks = (total_chunks - grp + C::WAVES_M - 1) / C::WAVES_M;   // chunks q = i*WAVES_M + grp
  float* myacc = accL + (size_t)grp * (T + 1) * C::LDC + n0 + 4 * g4;

  // Per-chunk metadata (kernel offset, 4 output rows, gather row of the chunk DDEPTH-1 ahead) is read from LDS
  // one chunk early, and the accumulator rows are read BEFORE the MFMA chain, so that the whole body has a single
  // LDS wait that overlaps the MFMAs (the first version had 4-5 serialised LDS round trips per chunk).
  // ---- operand ring.  One slot = the A fragment (16 gathered rows) AND the W[k] fragment of one chunk; the slot
  // of chunk i + DDEPTH - 1 is requested while chunk i is computed.  A and W of a chunk are requested together
  // because s_waitcnt vmcnt is IN ORDER: loads must be issued in the order they are consumed, otherwise waiting
  // for a late-issued/early-needed load (W one chunk ahead, as in an earlier version) drains the whole ring and
  // gather, W, MFMA and accumulate time add up instead of overlapping (measured: profiles/r01, tools/bench_sconv).
  // A rows come in through a buffer resource whose hardware bounds check returns 0 for the padding pairs
  // (row -1 -> offset beyond num_records): no predicate, no branch, counted waits.
  const __amdgpu_buffer_rsrc_t a_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (int)in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, (int)w_bytes, 0x00020000);
  using wfrag_t = std::conditional_t<BF16, uint2, f32x4>;      // one W fragment: 4 values per lane
  f32x4 aring[DDEPTH][C::KSTEPS];
  wfrag_t wring[DDEPTH][C::NT][C::KSTEPS];
  auto slot_load = [&](int32_t j, int k, auto RS) {
    constexpr int rs = decltype(RS)::value;
    // W is pre-packed in fragment order (pack_sconv_weights): one coalesced float4 per lane per (nt, t)
    constexpr uint32_t EB = BF16 ? 2u : 4u;                    // bytes per packed weight
    const uint32_t woff = ((uint32_t)k * (uint32_t)(CIN * COUT) + (uint32_t)(nsl * C::NT * C::KSTEPS * 64 + lane) * 4u) * EB;
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
      for (int t = 0; t < C::KSTEPS; ++t) {
        const int o = (int)(woff + (uint32_t)((nt * C::KSTEPS + t) * 64 * 4) * EB);
        if constexpr (BF16)
          wring[rs][nt][t] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(w_rsrc, o, 0, 0));
        else
          wring[rs][nt][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, o, 0, 0));
      }
    const uint32_t off = (uint32_t)j * (uint32_t)(CIN * 4) + (uint32_t)(16 * g4);   // j = -1 -> >= 2^32 - CIN*4
#pragma unroll
    for (int t = 0; t < C::KSTEPS; ++t)
      aring[rs][t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)(off + 64 * t), 0, 0));
  };
  auto chunk_of = [&](int i) { return min(i * C::WAVES_M + grp, total_chunks); };   // total_chunks = padding chunk

  // metadata: output rows of the chunk the next body() works on; gather row + offset of the chunk it prefetches
  // (operands are fed to the MFMA swapped — D^T = W^T A^T — so that lane (pair = l15, g) ends up with FOUR CONSECUTIVE
  //  COLUMNS 4g..4g+3 of its pair's output row: the accumulate is one 16-byte LDS read + one 16-byte write per column
  //  tile and the lane needs the row of one pair only; with the natural order it was four scattered 4-byte
  //  read-modify-writes plus the decode of four row indices — a third of the loop's instruction stream)
  int m_row = pr[chunk_of(0) * 16 + l15];
  int32_t m_j = pj[chunk_of(DDEPTH - 1) * 16 + l15];
  int m_k = ck[chunk_of(DDEPTH - 1)];

  auto body = [&](int i, auto RS) {
    constexpr int rs = decltype(RS)::value;
    float* const arow = myacc + m_row * C::LDC;
    slot_load(m_j, m_k, std::integral_constant<int, (rs + DDEPTH - 1) % DDEPTH>{});
    // LDS reads issued ahead of the MFMA chain: next chunk's metadata + this chunk's accumulator row
    m_row = pr[chunk_of(i + 1) * 16 + l15];
    const int qp = chunk_of(i + DDEPTH);
    m_j = pj[qp * 16 + l15];
    m_k = ck[qp];
    f32x4 old[C::NT];
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt) old[nt] = *reinterpret_cast<const f32x4*>(arow + nt * 16);
    f32x4 acc[C::NT];
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (BF16) {
#pragma unroll
      for (int t2 = 0; t2 < C::KSTEPS / 2; ++t2) {
        const f32x4 a0 = aring[rs][2 * t2], a1 = aring[rs][2 * t2 + 1];
        const uint4 ap = make_uint4(pack_bf16x2(a0[0], a0[1]), pack_bf16x2(a0[2], a0[3]), pack_bf16x2(a1[0], a1[1]),
                                    pack_bf16x2(a1[2], a1[3]));
        const bf16x8_t av = __builtin_bit_cast(bf16x8_t, ap);
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) {
          const uint2 w0 = wring[rs][nt][2 * t2], w1 = wring[rs][nt][2 * t2 + 1];
          const bf16x8_t bv = __builtin_bit_cast(bf16x8_t, make_uint4(w0.x, w0.y, w1.x, w1.y));
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv, av, acc[nt], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < C::KSTEPS; ++t) {
        const f32x4 a4 = aring[rs][t];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int nt = 0; nt < C::NT; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wring[rs][nt][t][u], a4[u], acc[nt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt) *reinterpret_cast<f32x4*>(arow + nt * 16) = old[nt] + acc[nt];
  };
  // prologue: fill the ring; main loop unrolled DDEPTH x so that ring slots are compile-time constants
  [&]<int... Is>(std::integer_sequence<int, Is...>) {
    (slot_load(pj[chunk_of(Is) * 16 + l15], ck[chunk_of(Is)], std::integral_constant<int, Is>{}), ...);
  }(std::make_integer_sequence<int, DDEPTH - 1>{});
  // Straight-line loop body: the trip count is rounded up to whole groups of DDEPTH chunks (the surplus chunks
  // are the all-padding chunk: zero A rows, +0 on the dummy accumulator row) and made provably wave-uniform, so
  // there is no branch between the loads and their waits — with per-chunk bounds checks hipcc emitted
  // s_waitcnt vmcnt(0) at the head of every group and the ring never overlapped anything.
  const int n_groups = (relu & 2) ? 0 : __builtin_amdgcn_readfirstlane((my_chunks + DDEPTH - 1) / DDEPTH);   // bit 1: measurement hook
  for (int g = 0; g < n_groups; ++g) {
    [&]<int... Is>(std::integer_sequence<int, Is...>) {
      (body(g * DDEPTH + Is, std::integral_constant<int, Is>{}), ...);
    }(std::make_integer_sequence<int, DDEPTH>{});
  }
  __syncthreads();

  // ---- epilogue: (split: raw partial sums) | BN scale/shift (+ReLU); one coalesced store per element
  constexpr int O4 = COUT / 4;
  float* dst = out + (nsplit > 1 ? (size_t)split * n_out * COUT : 0);
  for (int e = tid; e < rows * O4; e += 256) {
    const int r = e / O4, c4 = e - r * O4;
    float4 v = *reinterpret_cast<const float4*>(accL + r * C::LDC + c4 * 4);
#pragma unroll
    for (int g = 1; g < C::WAVES_M; ++g) {
      const float4 w = *reinterpret_cast<const float4*>(accL + ((size_t)g * (T + 1) + r) * C::LDC + c4 * 4);
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    if (nsplit == 1) {
      if (scale) {
        const float4 sc = reinterpret_cast<const float4*>(scale)[c4];
        const float4 sh = reinterpret_cast<const float4*>(shift)[c4];
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
      }
      if (relu & 1) {
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

// W[k][ci][co] (reference layout) -> fragment order Wp[k][nsl][nt][t][lane][u] =
//   W[k][16t + 4(lane>>4) + u][nsl*NW + nt*16 + (lane&15)], NW = 16 for cout <= 32 else 32
__global__ void pack_sconv_weights_kernel(const float* __restrict__ W, int K, int cin, int cout,
                                          float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per_k = (int64_t)cin * cout;
  if (e >= K * per_k) return;
  const int k = (int)(e / per_k);
  int64_t r = e - k * per_k;
  const int nw = cout <= 32 ? 16 : 32, ntn = nw / 16, ksteps = cin / 16;
  const int u = (int)(r & 3); r >>= 2;
  const int lane = (int)(r & 63); r >>= 6;
  const int t = (int)(r % ksteps); r /= ksteps;
  const int nt = (int)(r % ntn); r /= ntn;
  const int nsl = (int)r;
  const int ci = 16 * t + 4 * (lane >> 4) + u;
  const int co = nsl * nw + nt * 16 + (lane & 15);
  out[e] = W[(int64_t)k * per_k + (int64_t)ci * cout + co];
}
// same fragment order, values rounded to bf16 (round to nearest even): 2 bytes per weight
__global__ void pack_sconv_weights_bf16_kernel(const float* __restrict__ W, int K, int cin, int cout,
                                               uint16_t* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per_k = (int64_t)cin * cout;
  if (e >= K * per_k) return;
  const int k = (int)(e / per_k);
  int64_t r = e - k * per_k;
  const int nw = cout <= 32 ? 16 : 32, ntn = nw / 16, ksteps = cin / 16;
  const int u = (int)(r & 3); r >>= 2;
  const int lane = (int)(r & 63); r >>= 6;
  const int t = (int)(r % ksteps); r /= ksteps;
  const int nt = (int)(r % ntn); r /= ntn;
  const int nsl = (int)r;
  const int ci = 16 * t + 4 * (lane >> 4) + u;
  const int co = nsl * nw + nt * 16 + (lane & 15);
  uint32_t v = __float_as_uint(W[(int64_t)k * per_k + (int64_t)ci * cout + co]);
  v += 0x7FFFu + ((v >> 16) & 1u);
  out[e] = (uint16_t)(v >> 16);
}
int pack_sconv_weights_bf16(const float* W, int K, int cin, int cout, void* out, hipStream_t stream) {
  const int64_t n = (int64_t)K * cin * cout;
  hipLaunchKernelGGL(pack_sconv_weights_bf16_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, W, K, cin, cout,
                     reinterpret_cast<uint16_t*>(out));
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}
int pack_sconv_weights(const float* W, int K, int cin, int cout, float* out, hipStream_t stream) {
  const int64_t n = (int64_t)K * cin * cout;
  hipLaunchKernelGGL(pack_sconv_weights_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, W, K, cin, cout,
                     out);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

static int g_sconv_tile = 0;   // tuning hook: 0 = auto, else forced tile rows (64 / 128)
static int g_sconv_abl = 0;
static int g_sconv_skip = 0;   // measurement only: 1 = weight loads return zeros (no traffic), 2 = gathers return zeros
void sconv_set_skip(int m) { g_sconv_skip = m; }
static int g_sconv_split_target = 128;   // workgroups wanted per launch before kernel offsets are split
void sconv_set_variant(int v) { g_sconv_abl = v & 7; g_sconv_split_target = (v & 4) ? 1 : ((v & 2) ? 512 : ((v & 1) ? 256 : 128)); g_sconv_tile = ((v >> 8) & 3) == 1 ? 64 : ((v >> 8) & 3) == 2 ? 128 : ((v >> 8) & 3) == 3 ? 32 : 0; }

template <int CIN, int COUT, bool BF16>
static int launch_sconv(const float* in, int64_t n_in, const int32_t* nbr, const float* W, const float* scale,
                        const float* shift, int relu, float* out, int32_t n_out, int K, float* scratch,
                        size_t scratch_floats, hipStream_t stream) {
  EGONN_REQUIRE((uint64_t)n_in * CIN * 4 < (1ull << 32) - 4096, EGONN_ERR_INVALID,
                "sconv: input feature map of %lld rows exceeds the 4 GiB buffer-resource range", (long long)n_in);
  const uint32_t in_bytes = (uint32_t)((uint64_t)n_in * CIN * 4);
  const uint32_t w_bytes = (g_sconv_skip & 1) ? 0u : (uint32_t)((size_t)K * CIN * COUT * (BF16 ? 2 : 4));
  // 64-row tiles: the kernel is latency bound, more resident workgroups beat better chunk fill (tools/bench_sconv.py)
  const int T = g_sconv_tile ? g_sconv_tile : 64;
  const int tiles = (int)cdiv(n_out, T);
  // small levels: split the kernel offsets over workgroups until the chip is busy
  int nsplit = 1;
  if (scratch && tiles < g_sconv_split_target) {
    nsplit = (int)std::min<int64_t>(K, cdiv(g_sconv_split_target, tiles));
    while (nsplit > 1 && (size_t)nsplit * n_out * COUT > scratch_floats) --nsplit;
  }
  const size_t lds = sconv_lds_bytes<CIN, COUT>(K, T);
  static bool attr_done = false;
  if (!attr_done) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sconv_mfma_kernel<CIN, COUT, BF16>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  float* dst = nsplit > 1 ? scratch : out;
  hipEvent_t* pev = prof_kernel_events();
  if (pev[0]) {      // bench.py roofline leg: time exactly this dispatch
    hipExtLaunchKernelGGL((sconv_mfma_kernel<CIN, COUT, BF16>), dim3((unsigned)tiles, (unsigned)nsplit), dim3(256), lds,
                          stream, pev[0], pev[1], 0, in, nbr, W, scale, shift, ((relu ? 1 : 0) | ((g_sconv_skip & 4) ? 2 : 0)), dst, n_out, K, T,
                          (g_sconv_skip & 2) ? 0u : in_bytes, w_bytes);
    pev[0] = pev[1] = nullptr;
  } else {
    hipLaunchKernelGGL((sconv_mfma_kernel<CIN, COUT, BF16>), dim3((unsigned)tiles, (unsigned)nsplit), dim3(256), lds,
                       stream, in, nbr, W, scale, shift, ((relu ? 1 : 0) | ((g_sconv_skip & 4) ? 2 : 0)), dst, n_out, K, T,
                       (g_sconv_skip & 2) ? 0u : in_bytes, w_bytes);
  }
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

int sconv_forward(const float* in, int64_t n_in, const int32_t* nbr, const float* W, const float* Wp, const float* scale,
                  const float* shift, int relu, float* out, int32_t n_out, int K, int cin, int cout, float* scratch,
                  size_t scratch_floats, hipStream_t stream, int bf16) {
  if (n_out == 0) return EGONN_OK;
  EGONN_REQUIRE(K == 27 || K == 8, EGONN_ERR_INVALID, "sconv: kernel volume %d not supported", K);
  const bool mfma_shape = (cin == 32 && (cout == 32 || cout == 64)) || (cin == 64 && (cout == 64 || cout == 128)) ||
                          (cin == 128 && cout == 128) || (cin == 64 && cout == 32) || (cin == 128 && cout == 64);
  if (!g_force_naive && mfma_shape) {
    if (!Wp) {   // stand-alone operator call: pack into the tail of the scratch buffer
      const size_t wn = (size_t)K * cin * cout;
      EGONN_REQUIRE(scratch && scratch_floats > wn, EGONN_ERR_STATE, "sconv: no scratch to pack the kernel into");
      float* packed = scratch + (scratch_floats - wn);
      if (bf16)
        EGONN_TRY(pack_sconv_weights_bf16(W, K, cin, cout, packed, stream));
      else
        EGONN_TRY(pack_sconv_weights(W, K, cin, cout, packed, stream));
      Wp = packed;
      scratch_floats -= wn;
    }
    W = Wp;
#define EGONN_SCONV_CASE(CI, CO)                                                                                         \
  if (cin == CI && cout == CO)                                                                                           \
    return bf16 ? launch_sconv<CI, CO, true>(in, n_in, nbr, W, scale, shift, relu, out, n_out, K, scratch, scratch_floats,   \
                                             stream)                                                                     \
                : launch_sconv<CI, CO, false>(in, n_in, nbr, W, scale, shift, relu, out, n_out, K, scratch, scratch_floats,  \
                                              stream);
    EGONN_SCONV_CASE(32, 32)
    EGONN_SCONV_CASE(32, 64)
    EGONN_SCONV_CASE(64, 64)
    EGONN_SCONV_CASE(64, 128)
    EGONN_SCONV_CASE(128, 128)
    EGONN_SCONV_CASE(64, 32)      // input gradients of the 32->64 / 64->128 layers (training)
    EGONN_SCONV_CASE(128, 64)
#undef EGONN_SCONV_CASE
  }
  EGONN_REQUIRE(!bf16, EGONN_ERR_INVALID, "sconv: the bf16 operand path exists for the MFMA channel plans only (%d->%d)", cin, cout);
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

// UNIT: the (N,1) input features are all ones (what the reference always feeds: eval/evaluate.py:334,
// datasets/dataset_utils.py:80) -> an A-operand is just the occupancy bit, no rank/popcount, no feature gather.
template <bool UNIT>
__global__ __launch_bounds__(256) void conv0_k5_kernel(const float* __restrict__ feat,         // [n0] (Cin = 1)
                                                        const uint64_t* __restrict__ vkeys,     // level 0
                                                        const int32_t* __restrict__ g0,         // level-2 block of row
                                                        const uint64_t* __restrict__ t2m,       // [n2][27] masks
                                                        const int32_t* __restrict__ t2s,        // [n2][27] first rows
                                                        int32_t n2, int32_t nvox,
                                                        int32_t ntiles, const float* __restrict__ W,   // [125][32]
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int relu,
                                                        float* __restrict__ out,
                                                        unsigned long long* __restrict__ pair_counter,
                                                        const uint16_t* __restrict__ lut) {
  __shared__ uint64_t s_m[4][16][27];
  __shared__ int32_t s_s[4][16][27];
  // (local voxel position inside its 4x4x4 block, kernel offset) -> (adjacent-block slot << 6 | bit inside that
  // block's occupancy mask); 0xFFFF for the 3 padding offsets.  Built once per (persistent) workgroup.
  __shared__ uint16_t s_lut[64 * 128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g4 = lane >> 4;
  for (int e = tid; e < 64 * 128 / 2; e += 256)       // precomputed table (conv0_lut_host), 16 KB, coalesced
    reinterpret_cast<uint32_t*>(s_lut)[e] = reinterpret_cast<const uint32_t*>(lut)[e];
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
        (&s_m[wave][0][0])[idx] = pm[e];
        (&s_s[wave][0][0])[idx] = ps[e];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- prefetch: table of the next tile, block index + key bits of the one after
    table_issue(g_nxt);
    row_info(tile + 2 * tstep, g_nn, lk_nn);
    const uint32_t lk = lk_cur;
    const uint16_t* lrow = s_lut + lk * 128 + 4 * g4;
    float a[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) {
      const uint32_t en = lrow[16 * (q >> 2) + (q & 3)];
      float v = 0.f;
      if (en != 0xFFFFu) {
        const uint32_t slot = en >> 6, bit = en & 63;
        const uint64_t m = s_m[wave][l15][slot];          // rows beyond nvox have all-zero masks
        if ((m >> bit) & 1) {
          if constexpr (UNIT) v = 1.f;
          else v = feat[s_s[wave][l15][slot] + __popcll(m & ((1ull << bit) - 1))];
          npairs += (r0 + l15 < nvox);
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
    g_cur = g_nxt; lk_cur = lk_nxt;
    g_nxt = g_nn; lk_nxt = lk_nn;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) npairs += __shfl_xor(npairs, o, 64);
  __shared__ int32_t s_np[4];
  if (lane == 0) s_np[wave] = npairs;
  __syncthreads();
  // one atomic per workgroup, spread over 8 addresses (same-address atomics cost ~12 ns each: profiles/r01b)
  if (tid == 0)
    atomicAdd(pair_counter + 8 + (blockIdx.x & 7), (unsigned long long)(s_np[0] + s_np[1] + s_np[2] + s_np[3]));
}

// (local voxel position, kernel offset) -> (adjacent-block slot << 6 | bit in that block's mask), 0xFFFF = padding
static void conv0_lut_host(uint16_t* lut) {
  for (int lk = 0; lk < 64; ++lk)
    for (int k = 0; k < 128; ++k) {
      uint32_t v = 0xFFFFu;
      if (k < 125) {
        const int lx = (lk & 1) | ((lk >> 2) & 2), ly = ((lk >> 1) & 1) | ((lk >> 3) & 2), lz = ((lk >> 2) & 1) | ((lk >> 4) & 2);
        const int nx = lx + k % 5 - 2, ny = ly + (k / 5) % 5 - 2, nz = lz + k / 25 - 2;
        const int slot = ((nx < 0) ? 0 : (nx > 3 ? 2 : 1)) + 3 * ((ny < 0) ? 0 : (ny > 3 ? 2 : 1)) +
                         9 * ((nz < 0) ? 0 : (nz > 3 ? 2 : 1));
        const uint32_t ux = (uint32_t)nx & 3, uy = (uint32_t)ny & 3, uz = (uint32_t)nz & 3;
        const uint32_t bit = (ux & 1) | ((uy & 1) << 1) | ((uz & 1) << 2) | ((ux & 2) << 2) | ((uy & 2) << 3) | ((uz & 2) << 4);
        v = ((uint32_t)slot << 6) | bit;
      }
      lut[lk * 128 + k] = (uint16_t)v;
    }
}

int conv0_k5_forward(Ctx* ctx, const float* feat, const float* W, int cout, const float* scale,
                     const float* shift, int relu, float* out, hipStream_t stream) {
  const Plan& P = ctx->plan;
  if (!ctx->conv0_lut) {
    std::vector<uint16_t> h(64 * 128);
    conv0_lut_host(h.data());
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ctx->conv0_lut), h.size() * sizeof(uint16_t)));
    HIP_CHECK(hipMemcpy(ctx->conv0_lut, h.data(), h.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  }
  EGONN_REQUIRE(cout == COUT0, EGONN_ERR_INVALID, "conv0: %d output channels not supported (expected %d)", cout, COUT0);
  const Level& V = P.lv[0];
  const Level& B = P.lv[2];
  if (V.n == 0) return EGONN_OK;
  const int32_t ntiles = (int32_t)cdiv(V.n, 16);
  const unsigned grid = (unsigned)std::min<int64_t>(cdiv(ntiles, 4), 1536);
  EGONN_REQUIRE(P.g0 && P.t2m && P.t2s, EGONN_ERR_STATE, "conv0: plan has no block-neighbourhood table");
  if (feat)
    hipLaunchKernelGGL(conv0_k5_kernel<false>, dim3(grid), dim3(256), 0, stream, feat, V.keys, P.g0, P.t2m, P.t2s,
                       (int32_t)B.n, (int32_t)V.n, ntiles, W, scale, shift, relu, out, ctx->dev_pairs, ctx->conv0_lut);
  else   // unit features
    hipLaunchKernelGGL(conv0_k5_kernel<true>, dim3(grid), dim3(256), 0, stream, feat, V.keys, P.g0, P.t2m, P.t2s,
                       (int32_t)B.n, (int32_t)V.n, ntiles, W, scale, shift, relu, out, ctx->dev_pairs, ctx->conv0_lut);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

}  // namespace egonn
