// Sparse convolution for gfx950: row-group kernels with register accumulators.
//
// Three kernels share the decomposition below and produce bitwise-identical results:
//   sconv_rg_kernel   operands through a register ring            (small launches, bf16 maps)
//   sconv_dma_kernel  gathered rows by LDS-DMA in full 128-B lines (fp32 maps, launches that fill the chip)
//   sconv_wg_kernel   W slabs shared by a workgroup through LDS   (bf16 maps, big layers with >= 64 channels)
//
// Replaces MinkowskiConvolution / MinkowskiConvolutionTranspose forward as called from the reference
// (models/minkgl.py:39,100,105 and :46-60; ME BasicBlock conv1/conv2 via layers/eca_block.py:58-63;
// models/minkfpn.py:50-52):
//
//   out[o] = act( (sum_k in[nbr[o][k]] @ W[k]) * scale + shift )
//
// Work decomposition (tables: rowgroup.hip).  A wave owns one GROUP of 16 output rows x 32 output columns; its
// accumulators (2 MFMA tiles, 8 VGPRs) never leave registers.  It walks the kernel offsets k present in the group
// (bits of gmask[g], scalar loop) and the 32-channel blocks cb of the input; one ITEM (k, cb) is
//   fp32:  A = 16 gathered rows x 32 ch  (2 x buffer_load_b128 per lane)     W[k][cb][32 cols] (4 x b128, fragment order)
//          16 x v_mfma_f32_16x16x4_f32   (exact fp32)
//   bf16:  A = 16 rows x 32 ch           (1 x b128)                          W (2 x b128)
//           2 x v_mfma_f32_16x16x32_bf16 (fp32 accumulate)
// Missing neighbours are row "-1": the buffer resource's bounds check returns zeros without touching memory, so the
// loop has no predicates.  Operands are fed swapped (D^T = W^T A^T): lane (row = l&15, g = l>>4) ends up with four
// CONSECUTIVE output columns of its row, i.e. one 16-byte (fp32) / 8-byte (bf16) store per tile, BN scale/shift and
// ReLU fused.  There is no LDS accumulator, no atomics and no split over offsets: every output row is produced by one wave
// (or, for >= 64 input channels, by the KSP waves of one workgroup that split the channel blocks and add their partials
// in fixed order), summed in ascending k, ascending channel order => results do not depend on the grouping, the batch or
// the kernel variant (batch-invariant, bitwise reproducible).
// Items are software pipelined through a register ring (D slots of A+W), issued in consumption order because
// s_waitcnt vmcnt is in-order; the neighbour rows of a group (K x 16 ints) are staged once into wave-private LDS so the
// index lookups ride the lgkm counter and never drain the vector-memory queue.
// The number of groups is read from device memory (meta[0]); the grid is sized from the host's upper bound, one task per
// workgroup (workgroup = the KSP waves of a tile), and every XCD takes a contiguous eighth of the tasks (block b runs on
// XCD b % 8: consecutive groups — Z-order neighbours — share an L2).
// Optional epilogue: per-group column sums of the stored values (fixed order) for the ECA / GeM pooling.
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <utility>

#include "common.h"
#include <hip/hip_ext.h>
#include "kernels.h"

namespace egonn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8_t __attribute__((ext_vector_type(8)));

__device__ static inline uint32_t f2bf(float a) {       // round to nearest even
  uint32_t u = __float_as_uint(a);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}

// ------------------------------------------------------------------ weight packing
// W[k][ci][co] (reference layout: MinkowskiConvolution.kernel) -> item-major fragment order
//   fp32:  Wp[k][cb][ns][nt][t][lane][u] = W[k][32cb + 16t + 4(lane>>4) + u][32ns + 16nt + (lane&15)]
//   bf16:  Wp[k][cb][ns][nt][lane][e]    = W[k][32cb + 8(lane>>4) + e]      [32ns + 16nt + (lane&15)]   (rounded)
// so that the SLAB W[k][cb][all columns] the workgroup stages into LDS per item is one contiguous block.
// flip: the kernel of the input-gradient convolution, W'[k] = W[K-1-k]^T (same map, nbr[o][k]=j <=> nbr[j][K-1-k]=o);
// `transpose` alone serves k=2,s=2 <-> transposed pairs (same slot, W^T).
__global__ void pack_rg_weights_kernel(const float* __restrict__ W, int K, int cin, int cout, int bf16, int flip,
                                       int transpose, void* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per_k = (int64_t)cin * cout;
  if (e >= K * per_k) return;
  const int ncb = cin / 32, ns_n = cout / 32;
  int64_t r = e;
  int ci, co, k;
  if (!bf16) {
    const int u = (int)(r & 3); r >>= 2;
    const int lane = (int)(r & 63); r >>= 6;
    const int t = (int)(r & 1); r >>= 1;
    const int nt = (int)(r & 1); r >>= 1;
    const int ns = (int)(r % ns_n); r /= ns_n;
    const int cb = (int)(r % ncb); r /= ncb;
    k = (int)r;
    ci = 32 * cb + 16 * t + 4 * (lane >> 4) + u;
    co = 32 * ns + 16 * nt + (lane & 15);
  } else {
    const int el = (int)(r & 7); r >>= 3;
    const int lane = (int)(r & 63); r >>= 6;
    const int nt = (int)(r & 1); r >>= 1;
    const int ns = (int)(r % ns_n); r /= ns_n;
    const int cb = (int)(r % ncb); r /= ncb;
    k = (int)r;
    ci = 32 * cb + 8 * (lane >> 4) + el;
    co = 32 * ns + 16 * nt + (lane & 15);
  }
  const int ks = flip ? K - 1 - k : k;
  // source kernel is [K][cin_src][cout_src]; with transpose the packed (ci, co) reads W[ks][co][ci] of a [cout][cin] kernel
  const float v = transpose ? W[(int64_t)ks * per_k + (int64_t)co * cin + ci] : W[(int64_t)ks * per_k + (int64_t)ci * cout + co];
  if (bf16) reinterpret_cast<uint16_t*>(out)[e] = (uint16_t)f2bf(v);
  else reinterpret_cast<float*>(out)[e] = v;
}

int pack_rg_weights(const float* W, int K, int cin, int cout, int bf16, int flip, int transpose, void* out,
                    hipStream_t stream) {
  EGONN_REQUIRE(cin % 32 == 0 && cout % 32 == 0, EGONN_ERR_INVALID, "sconv: channel counts must be multiples of 32 (%d->%d)", cin, cout);
  const int64_t n = (int64_t)K * cin * cout;
  hipLaunchKernelGGL(pack_rg_weights_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, W, K, cin, cout, bf16, flip,
                     transpose, out);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------ the kernel
struct SconvArgs {
  const void* in;            // [n_in][CIN] fp32 or bf16
  const int32_t* snbr;       // row-group tables
  const uint32_t* gmask;
  const int32_t* perm;
  const int32_t* meta;       // [0] = groups in use
  const void* Wp;            // packed kernel (pack_rg_weights)
  const float* scale;        // folded BatchNorm (nullable)
  const float* shift;
  void* out;                 // [n_out][COUT]
  float* psum;               // [groups][COUT] column sums of the stored values (nullable)
  uint32_t in_bytes, w_bytes;
  int K, relu;
  int cap_groups = 0;                    // groups the tables hold: meta[0] is clipped to it (a batch beyond its reservation)
  const int32_t* order4 = nullptr;       // dispatch order of tasks of 4 consecutive groups (rowgroup.hip: longest first inside
                                         // every XCD's eighth); used by the kernels whose task is exactly such a quadruple
  unsigned long long* trace = nullptr;   // measurement builds only (tools/sconv_trace.py): 8 u64 per wave task
  int32_t* flags = nullptr;              // SPLIT instantiation: the plan's flag word (bit 3: fp16 range guard, sconv_split.hip)
  const float* res = nullptr;            // fp32 maps: out += res[row] in the epilogue (MinkHead's FPN step, models/minkgl.py:46-60)
};

// fp32 x 8 -> (hi, lo) fp16 x 8 (the two-way split of sconv_split.hip, same rounding)
typedef _Float16 rg_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 rg_f16x2 __attribute__((ext_vector_type(2)));
typedef float rg_f32x2 __attribute__((ext_vector_type(2)));
__device__ static inline void rg_split8h(const f32x4& a0, const f32x4& a1, rg_f16x8& hi, rg_f16x8& lo) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float x0 = q < 2 ? a0[2 * q] : a1[2 * q - 4], x1 = q < 2 ? a0[2 * q + 1] : a1[2 * q - 3];
    const rg_f16x2 h = __builtin_convertvector((rg_f32x2){x0, x1}, rg_f16x2);
    const rg_f32x2 hf = __builtin_convertvector(h, rg_f32x2);
    const rg_f16x2 l = __builtin_convertvector((rg_f32x2){x0 - hf[0], x1 - hf[1]}, rg_f16x2);
    hi[2 * q] = h[0]; hi[2 * q + 1] = h[1];
    lo[2 * q] = l[0]; lo[2 * q + 1] = l[1];
  }
}

// KSP > 1 (layers with >= 64 input channels): the KSP waves of a workgroup that share one (group, 32-column) tile each
// take 1/KSP of the input-channel blocks and the partial accumulators are summed through LDS in fixed order — shorter
// dependent item chains per wave and KSP x the waves in flight (the tail levels have a few hundred groups: without the
// split one wave per SIMD walked 40 dependent items while the chip idled).
// G > 1: a wave owns G CONSECUTIVE groups (sorted by mask => nearly the same offsets present) and walks the union of their
// offsets; the W fragments of an item are loaded ONCE and feed the MFMAs of all G groups.  With G = 1 four of the six
// vector loads of an item are W fragments and the CU's single texture-address path is ~75 % busy feeding four SIMDs
// (6 loads x 16 cycles x 4 waves per 512 MFMA cycles); G = 2 makes it 8 loads per 1024 MFMA cycles.  A group that lacks
// the offset skips the item's MFMAs (wave-uniform branch; its gather rows are "-1" and cost no memory traffic).  The sum
// order of every output row is unchanged (ascending k, ascending channel) => results are bitwise identical for every G.
// SPLIT (fp32 maps of levels 6-7, round 6): the item's arithmetic is the two-way fp16 split of sconv_split.hip — p.Wp is the
// pack_split_weights form, whose 4 KB item (hi nt0 | hi nt1 | lo nt0 | lo nt1 fragments) sits where the fp32 item sat, the gathered
// fragment is split in registers and six MFMAs of 16 cycles replace sixteen of 32: a wave's chain of 27 items was 5.8 of the launch's
// 14.5 us in exact-fp32 MFMAs.  Same partition and order of the sum as the exact instantiation; epilogue * 1 / (pack scale); range guard.
template <int CIN, int COUT, bool BF16, int D, int KSP, int G, int NW, bool SPLIT = false>
__global__ __launch_bounds__(NW * 64) void sconv_rg_kernel(const SconvArgs p) {
  static_assert(!SPLIT || !BF16, "split arithmetic is for fp32 maps");
  constexpr int NS = COUT / 32, NCB = CIN / 32;
  constexpr int NCBL = NCB / KSP;                        // channel blocks per wave
  constexpr int TPW = NW / KSP;                          // tiles per workgroup
  static_assert(NCB % KSP == 0 && NW % KSP == 0, "bad channel split");
  constexpr int ES = BF16 ? 2 : 4;                       // bytes per feature element
  constexpr int ALD = BF16 ? 1 : 2;                      // b128 loads of A per lane per item and group
  constexpr int WLD = BF16 ? 2 : 4;                      // b128 loads of W per lane per item
  constexpr uint32_t ITEM_BYTES = 32 * 32 * ES;
  constexpr int NPIECE = (G * 27 * 4 + 63) / 64;         // 16-byte pieces of the neighbour table per lane
  constexpr int LDSW = NPIECE * 64 * 4 + 16;             // ints of wave-private LDS (table + one all-absent row)
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, g4 = lane >> 4;
  const int K = p.K;
  int32_t* const ldsw = lds + wave * LDSW;
  f32x4* const red = reinterpret_cast<f32x4*>(lds + NW * LDSW);        // [2 parities][NW waves][G][2 tiles][64 lanes] (KSP > 1)

  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Wp), 0, (int)p.w_bytes, 0x00020000);

  const int ngroups = __builtin_amdgcn_readfirstlane(min(p.meta[0], p.cap_groups));   // a multiple of 16
  const int ntiles = (ngroups / G) * NS;
  const int ntask = (ntiles + TPW - 1) / TPW;            // workgroup tasks
  // every XCD (block b runs on XCD b % 8) takes one contiguous eighth of the tasks: its slice of the feature map
  // (Z-order => spatially compact) and the kernel fit its 4 MB L2.  Measured: interleaving 256-row chunks over the
  // XCDs instead costs 20-25 % (profiles/r02d_sconv.log) — the gathers miss L2 far more than the balance gains.
  const int xcd = blockIdx.x & 7, nper = gridDim.x >> 3; // gridDim.x is a multiple of 8
  const int cpx = (ntask + 7) >> 3;
  const int sub = wave % KSP;
  int par = 0;

  // ---- epilogue: BN scale/shift (+ReLU), one store per tile; optional per-group column sums
  auto epilogue = [&](const f32x4* acc, int g, int ns, int32_t row, const f32x4 (&sc)[2], const f32x4 (&sh)[2]) {
    float sums[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int c0 = ns * 32 + nt * 16 + 4 * g4;
      f32x4 v = acc[nt];
      if constexpr (SPLIT) {
        const f32x4 z = v - v;                               // NaN for a non-finite accumulator (range guard)
        if (__builtin_expect((z[0] + z[1]) + (z[2] + z[3]) != 0.f, 0) && p.flags) atomicOr(p.flags, 8);
        v = v * *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.Wp) + p.w_bytes);      // 1 / (pack scale)
      }
      if (p.scale) v = v * sc[nt] + sh[nt];
      if (p.relu) {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = fmaxf(v[u], 0.f);
      }
      if constexpr (!BF16) {
        if (p.res && row >= 0) v += *reinterpret_cast<const f32x4*>(p.res + (int64_t)row * COUT + c0);
      }
      if (row >= 0) {
        if constexpr (BF16) {
          uint2 o;
          o.x = f2bf(v[0]) | (f2bf(v[1]) << 16);
          o.y = f2bf(v[2]) | (f2bf(v[3]) << 16);
          *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + (int64_t)row * COUT + c0) = o;
        } else {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (int64_t)row * COUT + c0) = v;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) sums[nt][u] = row >= 0 ? v[u] : 0.f;
    }
    if (p.psum) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float sv = sums[nt][u];
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) sv += __shfl_xor(sv, o, 64);
          sums[nt][u] = sv;
        }
      if (l15 == 0) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          *reinterpret_cast<f32x4*>(p.psum + (int64_t)g * COUT + ns * 32 + nt * 16 + 4 * g4) =
              (f32x4){sums[nt][0], sums[nt][1], sums[nt][2], sums[nt][3]};
      }
    }
  };

  for (int lt = blockIdx.x >> 3; lt < cpx; lt += nper) {
    int task = xcd * cpx + lt;
    if constexpr (G == 4 && NS == 1 && TPW == 1) {       // a task is one quadruple of groups: longest tasks first
      if (p.order4 && task < ntask) task = __builtin_amdgcn_readfirstlane(p.order4[task]);
    }
    const int tile = task * TPW + wave / KSP;
    const int sg = tile / NS, ns = tile - sg * NS;
    const int g0 = sg * G;
    // Everything that does not depend on other loads is requested up front: the group masks, the groups' neighbour
    // rows, the output rows of the epilogue and the BatchNorm vectors (a small launch is a chain of memory latencies —
    // levels 5-7 ran 20 us per launch for < 2 us of MFMA work when these loads were issued one after the other)
    const bool tile_ok = task < ntask && tile < ntiles;
    uint32_t gm[G];
    int4 piece[NPIECE];
    int32_t orow[G];
    f32x4 bsc[2] = {{1.f, 1.f, 1.f, 1.f}, {1.f, 1.f, 1.f, 1.f}}, bsh[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < G; ++j) { gm[j] = 0; orow[j] = -1; }
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) piece[i] = make_int4(-1, -1, -1, -1);
    if (tile_ok) {
      const int4* src = reinterpret_cast<const int4*>(p.snbr + (int64_t)g0 * K * 16);
      const int n16 = G * K * 4;                         // 16-byte pieces
#pragma unroll
      for (int j = 0; j < G; ++j) gm[j] = p.gmask[g0 + j];
#pragma unroll
      for (int i = 0; i < NPIECE; ++i)
        if (lane + 64 * i < n16) piece[i] = src[lane + 64 * i];
      if (sub == 0) {
#pragma unroll
        for (int j = 0; j < G; ++j) orow[j] = p.perm[(int64_t)(g0 + j) * 16 + l15];
        if (p.scale) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            bsc[nt] = *reinterpret_cast<const f32x4*>(p.scale + ns * 32 + nt * 16 + 4 * g4);
            bsh[nt] = *reinterpret_cast<const f32x4*>(p.shift + ns * 32 + nt * 16 + 4 * g4);
          }
        }
      }
    }
    uint32_t un = 0;
#pragma unroll
    for (int j = 0; j < G; ++j) { gm[j] = __builtin_amdgcn_readfirstlane(gm[j]); un |= gm[j]; }
    const bool active = (un >> 31) != 0;
    if constexpr (KSP == 1) {
      if (!active) continue;                             // no barrier in this configuration: waves are independent
    }
    f32x4 acc[G][2];
#pragma unroll
    for (int j = 0; j < G; ++j) acc[j][0] = acc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (KSP == 1 || active) {
      // ---- the groups' neighbour rows (G x K x 16 ints, same order as in memory) + one all-absent row -> wave-private
      // LDS (an inactive group's rows are all absent)
      {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) reinterpret_cast<int4*>(ldsw)[lane + 64 * i] = piece[i];   // beyond n16: zeros
        if (lane < 4) reinterpret_cast<int4*>(ldsw)[NPIECE * 64 + lane] = make_int4(-1, -1, -1, -1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }

      // ---- item generator (scalar): set bits of the union mask x this wave's channel blocks
      uint32_t mk = un & 0x07FFFFFFu;
      const int n_items = __builtin_amdgcn_readfirstlane(__popc(mk) * NCBL);
      int gen_k = 0, gen_cb = 0;
      // pending item (the one whose loads are issued next)
      int32_t pend_idx[G];                               // raw LDS values: consumed one step later, so the read latency is hidden
      uint32_t pend_acb;
      uint32_t pend_woff, pend_wbad;
      int pend_bit;                                      // offset of the pending item (27 = none: no group has that bit)
      auto generate = [&]() {                            // branch-free: scalar selects only
        const bool need = (gen_cb == 0);
        const bool take = need && (mk != 0);
        const bool valid = !need || take;                // items in the middle of a k are always real
        gen_k = take ? __builtin_ctz(mk | 0x80000000u) : gen_k;
        mk = take ? (mk & (mk - 1)) : mk;
        const int cb = sub * NCBL + gen_cb;
#pragma unroll
        for (int j = 0; j < G; ++j) pend_idx[j] = ldsw[(valid ? (j * K + gen_k) * 16 : NPIECE * 256) + l15];
        pend_acb = (uint32_t)(cb * 32 * ES);
        pend_woff = (uint32_t)(((gen_k * NCB + cb) * NS + ns)) * ITEM_BYTES;
        pend_wbad = valid ? 0u : 0x80000000u;
        pend_bit = valid ? gen_k : 27;
        gen_cb = (valid && gen_cb + 1 < NCBL) ? gen_cb + 1 : 0;
      };

      f32x4 aring[D][G][ALD];
      f32x4 wring[D][WLD];
      int bitring[D];
      auto issue = [&](auto RS) {
        constexpr int rs = decltype(RS)::value;
        const int wv = (int)((uint32_t)(lane * 16) | pend_wbad);
        const int ws = __builtin_amdgcn_readfirstlane((int)pend_woff);
        bitring[rs] = pend_bit;
#pragma unroll
        for (int i = 0; i < WLD; ++i)
          wring[rs][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wv + i * 1024, ws, 0));
#pragma unroll
        for (int j = 0; j < G; ++j) {
          const uint32_t aoff = (uint32_t)pend_idx[j] * (uint32_t)(CIN * ES) + pend_acb + (uint32_t)(g4 * 16);
#pragma unroll
          for (int i = 0; i < ALD; ++i)
            aring[rs][j][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)(aoff + 64 * i), 0, 0));
        }
      };
      auto compute = [&](auto RS) {
        constexpr int rs = decltype(RS)::value;
#pragma unroll
        for (int j = 0; j < G; ++j) {
          if (G > 1 && !((gm[j] >> bitring[rs]) & 1u)) continue;        // wave-uniform
          if constexpr (SPLIT) {
            rg_f16x8 ah, al;
            rg_split8h(aring[rs][j][0], aring[rs][j][1], ah, al);
            const rg_f16x8 w0 = __builtin_bit_cast(rg_f16x8, wring[rs][0]), w1 = __builtin_bit_cast(rg_f16x8, wring[rs][1]);
            const rg_f16x8 w2 = __builtin_bit_cast(rg_f16x8, wring[rs][2]), w3 = __builtin_bit_cast(rg_f16x8, wring[rs][3]);
            acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2, ah, acc[j][0], 0, 0, 0);      // small terms first
            acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w3, ah, acc[j][1], 0, 0, 0);
            acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, al, acc[j][0], 0, 0, 0);
            acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, al, acc[j][1], 0, 0, 0);
            acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, ah, acc[j][0], 0, 0, 0);
            acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, ah, acc[j][1], 0, 0, 0);
          } else if constexpr (BF16) {
            const bf16x8_t av = __builtin_bit_cast(bf16x8_t, aring[rs][j][0]);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wring[rs][nt]), av, acc[j][nt], 0, 0, 0);
          } else {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
              for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                  acc[j][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wring[rs][nt * 2 + tt][u], aring[rs][j][tt][u], acc[j][nt], 0, 0, 0);
          }
        }
      };

      // ---- prologue: D-1 items in flight
      generate();
      [&]<int... Is>(std::integer_sequence<int, Is...>) {
        ((issue(std::integral_constant<int, Is>{}), generate(), __builtin_amdgcn_sched_barrier(0)), ...);
      }(std::make_integer_sequence<int, D - 1>{});
      // ---- main loop: straight-line groups of D items (no branch between a load and its wait)
      const int n_main = n_items / D;
      for (int it = 0; it < n_main; ++it) {
        [&]<int... Is>(std::integer_sequence<int, Is...>) {
          // the scheduling barriers pin the issue order: hipcc otherwise sinks the loads below the MFMAs and interleaves
          // the loads of different items, and the in-order vmcnt then waits for data that is not needed yet
          ((issue(std::integral_constant<int, (Is + D - 1) % D>{}), generate(), __builtin_amdgcn_sched_barrier(0),
            compute(std::integral_constant<int, Is>{}), __builtin_amdgcn_sched_barrier(0)), ...);
        }(std::make_integer_sequence<int, D>{});
      }
      // ---- remainder (< D items, already in flight in slots 0..rem-1)
      const int rem = n_items - n_main * D;
      [&]<int... Is>(std::integer_sequence<int, Is...>) {
        ((Is < rem ? compute(std::integral_constant<int, Is>{}) : (void)0), ...);
      }(std::make_integer_sequence<int, D - 1>{});
      __builtin_amdgcn_wave_barrier();                   // the next task overwrites the wave's LDS rows
      if constexpr (KSP == 1) {
#pragma unroll
        for (int j = 0; j < G; ++j)
          if (gm[j] >> 31) epilogue(acc[j], g0 + j, ns, orow[j], bsc, bsh);
      }
    }

    if constexpr (KSP > 1) {                             // fixed-order sum of the channel-split partials
      f32x4* r = red + ((par * NW + wave) * 2 * G) * 64 + lane;
#pragma unroll
      for (int j = 0; j < G; ++j) {
        r[(2 * j) * 64] = acc[j][0];
        r[(2 * j + 1) * 64] = acc[j][1];
      }
      __syncthreads();
      if (sub == 0) {
#pragma unroll
        for (int q = 1; q < KSP; ++q) {
          const f32x4* o = red + ((par * NW + wave + q) * 2 * G) * 64 + lane;
#pragma unroll
          for (int j = 0; j < G; ++j) {
            acc[j][0] += o[(2 * j) * 64];
            acc[j][1] += o[(2 * j + 1) * 64];
          }
        }
      }
      par ^= 1;
      if (active && sub == 0) {
#pragma unroll
        for (int j = 0; j < G; ++j)
          if (gm[j] >> 31) epilogue(acc[j], g0 + j, ns, orow[j], bsc, bsh);
      }
    }
  }
}

// ------------------------------------------------------------------ gather by LDS-DMA (fp32 maps)
// What the ablation of the kernel above measured (profiles/r02n_sconv_ablation.txt, L1 32->32 fp32, 65 us): with the
// MFMAs removed and every load hitting one cache line it still ran 57 us; W loads alone cost 15 us, the gathers alone
// 21-26 us, the same gathers issued as FULL 128-byte lines (8 lanes per row, 8 rows per instruction) 10 us; and the
// MFMAs alone ran at 2/3 of their rate because ~16 vector instructions of address arithmetic sat between every
// 16-MFMA block (tools/exp/mfma_mix.hip: vector instructions between MFMA blocks cost MFMA time, scalar ones do not).
// Here
//  * the gathered rows go global -> LDS with buffer_load_dwordx4 ... lds in full-line pieces (two 1 KB pieces per item:
//    rows 0-7 and 8-15; lane L fetches chunk (L&7)^(L>>3) of row L>>3, so the lane-linear LDS image is XOR-swizzled) and
//    the operand fragments come back with two conflict-free ds_read_b128 (lane (row, g): chunks g and 4+g of its row);
//  * the feature map is addressed as a STRUCTURED buffer (stride = row bytes): vindex = the neighbour row straight from
//    the table (-1 = absent -> out of range -> zeros, no memory traffic), voffset = the lane's constant chunk, soffset =
//    the channel block: the texture-address unit does the multiply-add, the loop has no vector address arithmetic;
//  * the ring of gathered items lives in LDS, not in VGPRs; W fragments stay in a register ring.
// The compiler does not know that an LDS-DMA write feeds a later ds_read, and would drain the whole ring (vmcnt(0)) in
// front of any LDS read it can see; every LDS read inside the item loop is therefore issued from one asm statement per
// item that carries its own counted s_waitcnt vmcnt and its own lgkmcnt(0).
// Arithmetic (MFMA order, channel permutation, W packing, epilogue) is identical to sconv_rg_kernel => bitwise equal.
__device__ static inline float row16_sum(float v) {      // sum over the 16 lanes of a DPP row (= the 16 rows of a tile)
  int x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));   // row_half_mirror
  x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}

template <int CIN, int COUT, int D, int KSP, int NW, bool TRACE = false, bool PIPE = false>
__global__ __launch_bounds__(NW * 64) void sconv_dma_kernel(const SconvArgs p) {
  constexpr int NS = COUT / 32, NCB = CIN / 32;
  constexpr int NCBL = NCB / KSP;
  constexpr int TPW = NW / KSP;                          // tiles per workgroup
  static_assert(NCB % KSP == 0 && NW % KSP == 0, "bad channel split");
  constexpr uint32_t ITEM_BYTES = 32 * 32 * 4;
  constexpr int TBL = 2 * 256 + 32;                      // ints: 27 x 16 neighbour rows (padded to 512) + an all-absent row; 128-byte multiple
  constexpr int SLOT = 2048;                             // one gathered item: 16 rows x 128 B
  constexpr int WAVE_LDS = TBL * 4 + D * SLOT;
  constexpr int VM_PER_ITEM = 6;                         // 4 W loads + 2 LDS-DMA pieces
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, g4 = lane >> 4;
  const int K = p.K;
  char* const wl = smem + wave * WAVE_LDS;
  int32_t* const ldsw = reinterpret_cast<int32_t*>(wl);
  char* const ring = wl + TBL * 4;
  typedef __attribute__((address_space(3))) char lds_char;
  const uint32_t wl_addr = (uint32_t)(uintptr_t)(lds_char*)wl;          // LDS byte address of this wave's region
  // operand read addresses inside a slot: chunks g and 4+g of row l15 (XOR swizzle, see above)
  const uint32_t rd0 = wl_addr + TBL * 4 + (uint32_t)(l15 * 128 + ((g4 ^ (l15 & 7)) * 16));
  const uint32_t rd1 = wl_addr + TBL * 4 + (uint32_t)(l15 * 128 + (((4 + g4) ^ (l15 & 7)) * 16));
  // neighbour-table read address of this lane's DMA rows (rows L>>3 and 8 + (L>>3))
  const uint32_t tb0 = wl_addr + (uint32_t)((lane >> 3) * 4);
  const int dma_chunk = ((lane & 7) ^ (lane >> 3)) * 16;
  const int w_lane = lane * 16;
  const int w_oob = (int)(0x80000000u | (uint32_t)(lane * 16));   // items past the end: out-of-range W loads (no traffic)

  const __amdgpu_buffer_rsrc_t a_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), (short)(CIN * 4), (int)(p.in_bytes / (CIN * 4)), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Wp), 0, (int)p.w_bytes, 0x00020000);

  const int ngroups = __builtin_amdgcn_readfirstlane(min(p.meta[0], p.cap_groups));
  const int ntiles = ngroups * NS;
  const int ntask = (ntiles + TPW - 1) / TPW;
  // one workgroup per task (the hardware dispatcher balances the uneven tasks), contiguous eighth of the tasks per XCD.
  // A PERSISTENT variant of this kernel (resident workgroups pulling tiles from per-XCD atomic counters, next tables
  // prefetched, no relaunch) was built and measured: 111 us vs 62 us on L1 32->32 (profiles/r02n_sconv_ablation.txt) —
  // with all 16 wave slots of a CU live the per-item time rose from 1400 to 1870 cycles: the CU's vector-memory path is the
  // shared bottleneck and more resident waves only deepen its queue and thrash the 32 KB L1 with W fragments.
  const int xcd = blockIdx.x & 7, nper = gridDim.x >> 3;
  const int cpx = (ntask + 7) >> 3;
  const int sub = wave % KSP;

  auto epilogue = [&](const f32x4* acc, int g, int ns, int32_t row) {
    float sums[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int c0 = ns * 32 + nt * 16 + 4 * g4;
      f32x4 v = acc[nt];
      if (p.scale) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + c0);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + c0);
        v = v * sc + sh;
      }
      if (p.relu) {
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = fmaxf(v[u], 0.f);
      }
      if (row >= 0) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (int64_t)row * COUT + c0) = v;
#pragma unroll
      for (int u = 0; u < 4; ++u) sums[nt][u] = row >= 0 ? v[u] : 0.f;
    }
    if (p.psum) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int u = 0; u < 4; ++u) sums[nt][u] = row16_sum(sums[nt][u]);
      if (l15 == 0) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          *reinterpret_cast<f32x4*>(p.psum + (int64_t)g * COUT + ns * 32 + nt * 16 + 4 * g4) =
              (f32x4){sums[nt][0], sums[nt][1], sums[nt][2], sums[nt][3]};
      }
    }
  };

  for (int lt = blockIdx.x >> 3; lt < cpx; lt += nper) {
    const int task = xcd * cpx + lt;
    const int tile = task * TPW + wave / KSP;
    const int g = tile / NS, ns = tile - g * NS;
    const bool tile_ok = task < ntask && tile < ntiles;
    uint32_t gm = 0;
    int4 v0 = make_int4(-1, -1, -1, -1), v1 = make_int4(-1, -1, -1, -1);
    int32_t orow = -1;
    if (tile_ok) {
      const int4* src = reinterpret_cast<const int4*>(p.snbr + (int64_t)g * K * 16);
      const int n16 = K * 4;
      gm = p.gmask[g];
      if (lane < n16) v0 = src[lane];
      if (lane + 64 < n16) v1 = src[lane + 64];
      if (sub == 0) orow = p.perm[(int64_t)g * 16 + l15];
    }
    unsigned long long tr[5];
    if constexpr (TRACE) tr[0] = __builtin_amdgcn_s_memtime();
    gm = __builtin_amdgcn_readfirstlane(gm);
    const bool active = (gm >> 31) != 0;
    if constexpr (KSP == 1) {
      if (!active) continue;
    }
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};

    if (active) {
      {
        reinterpret_cast<int4*>(ldsw)[lane] = v0;
        reinterpret_cast<int4*>(ldsw)[lane + 64] = v1;   // pieces >= K*4: absent
        if (lane < 8) reinterpret_cast<int4*>(ldsw)[128 + lane] = make_int4(-1, -1, -1, -1);   // the all-absent row (512..)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      if constexpr (TRACE) tr[1] = __builtin_amdgcn_s_memtime();

      uint32_t mk = gm & 0x07FFFFFFu;
      const int n_items = __builtin_amdgcn_readfirstlane(__popc(mk) * NCBL);
      int gen_k = 0, gen_cb = 0;
      // the item whose loads are issued next: its two neighbour rows (written by the asm below), channel block, W offset
      int32_t pend_i0 = -1, pend_i1 = -1;
      int pend_acb = 0, pend_woff = 0;                  // pend_woff < 0: item past the end
      // the item after that (generated: table address known, rows not read yet)
      uint32_t nxt_tb = tb0 + 512 * 4;
      int nxt_acb = 0, nxt_woff = 0;
      auto generate = [&]() {                            // scalar selects only; fills nxt_*
        const bool need = (gen_cb == 0);
        const bool take = need && (mk != 0);
        const bool valid = !need || take;                // items past the end gather the all-absent row and out-of-range W: never used, no traffic
        gen_k = take ? __builtin_ctz(mk | 0x80000000u) : gen_k;
        mk = take ? (mk & (mk - 1)) : mk;
        const int cb = sub * NCBL + gen_cb;
        nxt_tb = tb0 + (uint32_t)((valid ? gen_k * 16 : 512) * 4);
        nxt_acb = cb * 128;
        nxt_woff = valid ? (int)((uint32_t)(((gen_k * NCB + cb) * NS + ns)) * ITEM_BYTES) : -1;
        gen_cb = (valid && gen_cb + 1 < NCBL) ? gen_cb + 1 : 0;
      };

      f32x4 wring[D][4];
      auto issue = [&](auto RS) {                        // loads of the pending item; then nxt -> pend (rows arrive by asm)
        constexpr int rs = decltype(RS)::value;
        const int wv = pend_woff >= 0 ? w_lane : w_oob;  // (scalar condition: one v_cndmask)
        const int ws = pend_woff >= 0 ? pend_woff : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          wring[rs][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wv + i * 1024, ws, 0));
        __builtin_amdgcn_struct_ptr_buffer_load_lds(a_rsrc, (lds_char*)(ring + rs * SLOT), 16, pend_i0, dma_chunk, pend_acb, 0, 0);
        __builtin_amdgcn_struct_ptr_buffer_load_lds(a_rsrc, (lds_char*)(ring + rs * SLOT + 1024), 16, pend_i1, dma_chunk, pend_acb, 0, 0);
      };
      // one asm per item: wait for the item's DMA pieces, read its operand fragments and the NEXT-to-issue item's rows
      auto fetch = [&](auto RS, auto VM, f32x4& a0, f32x4& a1) {
        constexpr int rs = decltype(RS)::value;
        constexpr int vm = decltype(VM)::value;
        const uint32_t r0 = rd0, r1 = rd1, tb = nxt_tb;  // (asm operands cannot name captures directly)
        int32_t i0, i1;
        asm volatile(
            "s_waitcnt vmcnt(%6)\n\t"
            "ds_read_b128 %0, %4 offset:%7\n\t"
            "ds_read_b128 %1, %5 offset:%7\n\t"
            "ds_read_b32 %2, %8\n\t"
            "ds_read_b32 %3, %8 offset:32\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(a0), "=&v"(a1), "=&v"(i0), "=&v"(i1)
            : "v"(r0), "v"(r1), "n"(vm), "n"(rs * SLOT), "v"(tb)
            : "memory");
        pend_i0 = i0; pend_i1 = i1;
        pend_acb = nxt_acb; pend_woff = nxt_woff;
      };
      auto mfmas = [&](auto RS, const f32x4& a0, const f32x4& a1) {
        constexpr int rs = decltype(RS)::value;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wring[rs][nt * 2 + tt][u], (tt ? a1 : a0)[u], acc[nt], 0, 0, 0);
      };

      // split-phase fetch (PIPE): the LDS reads of item i+1 are issued BEFORE the MFMAs of item i and awaited after them, so
      // the LDS round trip leaves the wave's per-item critical path (costs one ring slot of look-ahead: D = 4)
      auto fetch_issue = [&](auto RS, auto VM, f32x4& b0, f32x4& b1, int32_t& i0, int32_t& i1) {
        constexpr int rs = decltype(RS)::value;
        constexpr int vm = decltype(VM)::value;
        const uint32_t r0 = rd0, r1 = rd1, tb = nxt_tb;
        asm volatile(
            "s_waitcnt vmcnt(%6)\n\t"
            "ds_read_b128 %0, %4 offset:%7\n\t"
            "ds_read_b128 %1, %5 offset:%7\n\t"
            "ds_read_b32 %2, %8\n\t"
            "ds_read_b32 %3, %8 offset:32"
            : "=&v"(b0), "=&v"(b1), "=&v"(i0), "=&v"(i1)
            : "v"(r0), "v"(r1), "n"(vm), "n"(rs * SLOT), "v"(tb)
            : "memory");
      };
      auto fetch_wait = [&](f32x4& b0, f32x4& b1, int32_t& i0, int32_t& i1) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0), "+v"(b1), "+v"(i0), "+v"(i1)::"memory");
        pend_i0 = i0; pend_i1 = i1;
        pend_acb = nxt_acb; pend_woff = nxt_woff;
      };

      // ---- prologue: D-1 items in flight.  The neighbour rows of the first D items are read with ONE LDS round trip
      // (they were D dependent round trips: generate -> read -> issue -> generate -> read ...)
      {
        uint32_t ptb[D];
        int pacb[D], pwoff[D];
        int32_t pi0[D], pi1[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
          generate();
          ptb[j] = nxt_tb; pacb[j] = nxt_acb; pwoff[j] = nxt_woff;
        }
        static_assert(D == 3 || D == 4, "prologue asm is written for 3 or 4 ring slots");
        if constexpr (D == 4) {
          asm volatile(
              "ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:32\n\t"
              "ds_read_b32 %2, %9\n\tds_read_b32 %3, %9 offset:32\n\t"
              "ds_read_b32 %4, %10\n\tds_read_b32 %5, %10 offset:32\n\t"
              "ds_read_b32 %6, %11\n\tds_read_b32 %7, %11 offset:32\n\t"
              "s_waitcnt lgkmcnt(0)"
              : "=&v"(pi0[0]), "=&v"(pi1[0]), "=&v"(pi0[1]), "=&v"(pi1[1]), "=&v"(pi0[2]), "=&v"(pi1[2]), "=&v"(pi0[3]), "=&v"(pi1[3])
              : "v"(ptb[0]), "v"(ptb[1]), "v"(ptb[2]), "v"(ptb[3])
              : "memory");
        } else {
          asm volatile(
              "ds_read_b32 %0, %6\n\tds_read_b32 %1, %6 offset:32\n\t"
              "ds_read_b32 %2, %7\n\tds_read_b32 %3, %7 offset:32\n\t"
              "ds_read_b32 %4, %8\n\tds_read_b32 %5, %8 offset:32\n\t"
              "s_waitcnt lgkmcnt(0)"
              : "=&v"(pi0[0]), "=&v"(pi1[0]), "=&v"(pi0[1]), "=&v"(pi1[1]), "=&v"(pi0[2]), "=&v"(pi1[2])
              : "v"(ptb[0]), "v"(ptb[1]), "v"(ptb[2])
              : "memory");
        }
        [&]<int... Is>(std::integer_sequence<int, Is...>) {
          ((pend_i0 = pi0[Is], pend_i1 = pi1[Is], pend_acb = pacb[Is], pend_woff = pwoff[Is],
            issue(std::integral_constant<int, Is>{}), __builtin_amdgcn_sched_barrier(0)), ...);
        }(std::make_integer_sequence<int, D - 1>{});
        pend_i0 = pi0[D - 1]; pend_i1 = pi1[D - 1]; pend_acb = pacb[D - 1]; pend_woff = pwoff[D - 1];
      }
      if constexpr (TRACE) tr[2] = __builtin_amdgcn_s_memtime();
      const int n_main = n_items / D;
      const int rem = n_items - n_main * D;
      if constexpr (!PIPE) {
        // ---- main loop
        for (int it = 0; it < n_main; ++it) {
          [&]<int... Is>(std::integer_sequence<int, Is...>) {
            (([&] {
               f32x4 a0, a1;
               issue(std::integral_constant<int, (Is + D - 1) % D>{});
               generate();
               __builtin_amdgcn_sched_barrier(0);
               fetch(std::integral_constant<int, Is>{}, std::integral_constant<int, VM_PER_ITEM*(D - 1)>{}, a0, a1);
               mfmas(std::integral_constant<int, Is>{}, a0, a1);
               __builtin_amdgcn_sched_barrier(0);
             }()), ...);
          }(std::make_integer_sequence<int, D>{});
        }
        // ---- remainder (< D items, in flight in slots 0..rem-1; nothing more is issued: wait for everything once)
        [&]<int... Is>(std::integer_sequence<int, Is...>) {
          (([&] {
             if (Is < rem) {
               f32x4 a0, a1;
               fetch(std::integral_constant<int, Is>{}, std::integral_constant<int, 0>{}, a0, a1);
               mfmas(std::integral_constant<int, Is>{}, a0, a1);
             }
           }()), ...);
        }(std::make_integer_sequence<int, D - 1>{});
      } else {
        f32x4 a0, a1;
        {                                                  // operands of item 0 (its rows were read by rows_only above)
          const uint32_t r0 = rd0, r1 = rd1;
          asm volatile(
              "s_waitcnt vmcnt(%4)\n\t"
              "ds_read_b128 %0, %2\n\t"
              "ds_read_b128 %1, %3\n\t"
              "s_waitcnt lgkmcnt(0)"
              : "=&v"(a0), "=&v"(a1)
              : "v"(r0), "v"(r1), "n"(VM_PER_ITEM * (D - 2))
              : "memory");
        }
        for (int it = 0; it < n_main; ++it) {
          [&]<int... Is>(std::integer_sequence<int, Is...>) {
            (([&] {
               f32x4 b0, b1;
               int32_t i0, i1;
               issue(std::integral_constant<int, (Is + D - 1) % D>{});
               generate();
               __builtin_amdgcn_sched_barrier(0);
               fetch_issue(std::integral_constant<int, (Is + 1) % D>{}, std::integral_constant<int, VM_PER_ITEM*(D - 2)>{}, b0, b1, i0, i1);
               mfmas(std::integral_constant<int, Is>{}, a0, a1);
               fetch_wait(b0, b1, i0, i1);
               a0 = b0; a1 = b1;
               __builtin_amdgcn_sched_barrier(0);
             }()), ...);
          }(std::make_integer_sequence<int, D>{});
        }
        // ---- remainder: item n_main*D (slot 0) is already in (a0, a1); the others are fetched with everything landed
        [&]<int... Is>(std::integer_sequence<int, Is...>) {
          (([&] {
             if (Is < rem) {
               if constexpr (Is > 0) fetch(std::integral_constant<int, Is>{}, std::integral_constant<int, 0>{}, a0, a1);
               mfmas(std::integral_constant<int, Is>{}, a0, a1);
             }
           }()), ...);
        }(std::make_integer_sequence<int, D - 1>{});
      }
      if constexpr (TRACE) tr[3] = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // items past the end are still landing in the ring
      __builtin_amdgcn_wave_barrier();
      if constexpr (KSP == 1) {
        if (active) epilogue(acc, g, ns, orow);
      }
      if constexpr (TRACE) {
        tr[4] = __builtin_amdgcn_s_memtime();
        if (lane == 0 && p.trace) {
          unsigned long long* o = p.trace + ((int64_t)tile * KSP + sub) * 8;
          uint32_t hwid;
          asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
          o[0] = tr[0]; o[1] = tr[1]; o[2] = tr[2]; o[3] = tr[3]; o[4] = tr[4];
          o[5] = (unsigned long long)n_items; o[6] = hwid; o[7] = blockIdx.x;
        }
      }
    }

    if constexpr (KSP > 1) {                             // fixed-order sum of the channel-split partials (ring is drained)
      f32x4* r = reinterpret_cast<f32x4*>(ring) + lane;
      r[0] = acc[0];
      r[64] = acc[1];
      __syncthreads();
      if (sub == 0) {
#pragma unroll
        for (int q = 1; q < KSP; ++q) {
          const f32x4* o = reinterpret_cast<const f32x4*>(ring + q * WAVE_LDS) + lane;
          acc[0] += o[0];
          acc[1] += o[64];
        }
      }
      __syncthreads();                                   // the partner waves' next task writes into these slots
      if (active && sub == 0) epilogue(acc, g, ns, orow);
    }
  }
}

// NW waves per workgroup.  A workgroup keeps ALL its wave slots until its slowest wave has retired (and a retired wave keeps
// its slot until its last stores are acknowledged): with independent tiles of 2-27 items in one 4-wave workgroup a CU held
// 9 live waves of 16 (tools/sconv_trace.py).  The waves of a workgroup only have to be together when they split the input
// channels of one tile (KSP > 1), so a workgroup is exactly those KSP waves; the dispatcher starts > 3000 workgroups/us
// (tools/exp/dispatch_rate.hip), far more than these launches need.
template <int CIN, int COUT, int KSP, int D, bool TRACE = false, bool PIPE = false>
static int launch_dma_d(const SconvArgs& a, int64_t groups_hint, hipStream_t stream, int nw_sel = 0) {
  constexpr int NS = COUT / 32;
  auto go = [&](auto NWC) -> int {
    constexpr int NW = decltype(NWC)::value;
    const size_t lds = NW * ((2 * 256 + 32) * 4 + D * 2048);
    static AttrOnce attr_done;                       // per instantiation (the lambda is instantiated per NW)
    if (attr_done.need()) {
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sconv_dma_kernel<CIN, COUT, D, KSP, NW, TRACE, PIPE>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_done.mark(); 
    }
    const int64_t ntask = cdiv(groups_hint * NS * KSP, NW);
    int64_t grid = std::min<int64_t>(std::max<int64_t>(ntask, 8), 65536);
    grid = (grid + 7) / 8 * 8;
    hipEvent_t* pev = prof_kernel_events();
    if (pev[0]) {
      hipExtLaunchKernelGGL((sconv_dma_kernel<CIN, COUT, D, KSP, NW, TRACE, PIPE>), dim3((unsigned)grid), dim3(NW * 64), lds, stream, pev[0], pev[1], 0, a);
      pev[0] = pev[1] = nullptr;
    } else {
      hipLaunchKernelGGL((sconv_dma_kernel<CIN, COUT, D, KSP, NW, TRACE, PIPE>), dim3((unsigned)grid), dim3(NW * 64), lds, stream, a);
    }
    HIP_CHECK(hipGetLastError());
    return EGONN_OK;
  };
  if constexpr (KSP < 4) {
    if (nw_sel == 4) return go(std::integral_constant<int, 4>{});
  }
  return go(std::integral_constant<int, KSP>{});
}

// ------------------------------------------------------------------ workgroup-cooperative variant
// What the per-wave kernel above taught (profiles/r02a_pmc_sweep.txt): the texture-address unit is 71 % busy and the MFMA
// pipe 45-50 % — four of the six vector loads of an item are the W fragments, re-fetched by every wave for every item.
// Here the NW waves of a workgroup own NW consecutive groups (sorted => nearly the same offsets present) and walk the
// UNION of their offsets in lock-step; per item the slab W[k][cb][all COUT columns] is copied ONCE per workgroup into an
// LDS double buffer (each wave copies 1/NW of it) and every wave reads its B fragments from LDS (ds_read_b128,
// lane-linear, conflict-free).  A wave now covers ALL output columns of its group (accumulators: COUT/16 MFMA tiles),
// so the gathered rows are loaded once instead of once per 32-column slice; a wave whose group lacks the offset skips
// the item's MFMAs (wave-uniform branch) but still takes part in the copy and the barrier.  One barrier per item:
//   step i:  ds_write W(i+1) -> slab[(i+1)&1]   (data loaded one step earlier; slab last read in step i-1, before the barrier)
//            issue W(i+2) -> registers, A(i+2) -> ring slot            (issue order = consumption order: vmcnt is in-order)
//            if (group has k_i)  B <- slab[i&1];  MFMAs with A(i)
//            barrier
template <int CIN, int COUT, bool BF16, int NW>
__global__ __launch_bounds__(NW * 64) void sconv_wg_kernel(const SconvArgs p) {
  constexpr int D = 3;
  constexpr int NS = COUT / 32, NCB = CIN / 32;
  constexpr int ES = BF16 ? 2 : 4;
  constexpr int ALD = BF16 ? 1 : 2;                      // b128 loads of A per lane per item
  constexpr int BLD = BF16 ? 2 : 4;                      // ds_read_b128 of B fragments per 32-column slice
  constexpr uint32_t ITEM_BYTES = 32 * 32 * ES;          // one (k, cb, 32-column slice) fragment block
  constexpr uint32_t SLAB = ITEM_BYTES * NS;             // W[k][cb][all columns]
  constexpr int LB = SLAB / (NW * 64);                   // bytes of the slab each lane copies
  constexpr int SVEC = LB >= 16 ? 16 : 8;
  constexpr int SW = LB / SVEC;                          // stage loads per lane per item
  static_assert(LB >= 8 && SW * SVEC == LB, "slab / workgroup size mismatch");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g4 = lane >> 4;
  const int K = p.K;
  int32_t* const ldsw = reinterpret_cast<int32_t*>(smem + 2 * SLAB) + wave * (28 * 16);

  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Wp), 0, (int)p.w_bytes, 0x00020000);

  const int ngroups = __builtin_amdgcn_readfirstlane(min(p.meta[0], p.cap_groups));
  const int ntask = ngroups / NW;                        // groups in use are a multiple of 16
  const int xcd = blockIdx.x & 7, nper = gridDim.x >> 3; // gridDim.x is a multiple of 8; one contiguous eighth per XCD
  const int cpx = (ntask + 7) >> 3;

  for (int lt = blockIdx.x >> 3; lt < cpx; lt += nper) {
    int task = xcd * cpx + lt;
    if (task >= ntask) continue;
    if constexpr (NW == 4) {                             // longest tasks first (rowgroup_order_kernel)
      if (p.order4) task = __builtin_amdgcn_readfirstlane(p.order4[task]);
    }
    const int g0 = task * NW;
    uint32_t U = 0, own = 0;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const uint32_t m = p.gmask[g0 + j];
      U |= m;
      own = (j == wave) ? m : own;
    }
    U = __builtin_amdgcn_readfirstlane(U);
    own = __builtin_amdgcn_readfirstlane(own);
    if (!(U >> 31)) continue;                            // four empty groups (tail of a partial window)
    const int g = g0 + wave;

    // ---- the group's neighbour rows (K x 16 ints) + an all-absent row -> wave-private LDS
    {
      const int4* src = reinterpret_cast<const int4*>(p.snbr + (int64_t)g * K * 16);
      const int n16 = (own >> 31) ? K * 4 : 0;           // 16-byte pieces; an empty group reads nothing
      int4 v0 = make_int4(-1, -1, -1, -1), v1 = make_int4(-1, -1, -1, -1);
      if (lane < n16) v0 = src[lane];
      if (lane + 64 < n16) v1 = src[lane + 64];
      reinterpret_cast<int4*>(ldsw)[lane] = v0;
      if (lane + 64 < 28 * 4) reinterpret_cast<int4*>(ldsw)[lane + 64] = v1;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // ---- item generator (scalar, identical in every wave): set bits of the union mask x channel blocks
    uint32_t mk = U & 0x07FFFFFFu;
    const int n_items = __popc(mk) * NCB;
    const int n_steps = (n_items + D - 1) / D;           // groups of D steps; surplus steps are empty items
    int gen_k = 0, gen_cb = 0;
    int32_t pend_idx;
    uint32_t pend_acb, pend_woff, pend_wbad;
    int pend_has;
    auto generate = [&]() {                              // branch-free: scalar selects only
      const bool need = (gen_cb == 0);
      const bool take = need && (mk != 0);
      const bool valid = !need || take;
      gen_k = take ? __builtin_ctz(mk | 0x80000000u) : gen_k;
      mk = take ? (mk & (mk - 1)) : mk;
      pend_has = (valid && ((own >> gen_k) & 1u)) ? 1 : 0;
      const int krow = pend_has ? gen_k : K;
      pend_idx = ldsw[krow * 16 + l15];
      pend_acb = (uint32_t)(gen_cb * 32 * ES);
      pend_woff = (uint32_t)(gen_k * NCB + gen_cb) * SLAB;
      pend_wbad = valid ? 0u : 0x80000000u;
      gen_cb = (valid && gen_cb + 1 < NCB) ? gen_cb + 1 : 0;
    };

    using svec_t = std::conditional_t<SVEC == 16, f32x4, float2>;
    svec_t wreg[SW];
    f32x4 aring[D][ALD];
    int has[D];
    auto issue = [&](auto RS) {                          // W slab piece of the pending item, then its gathered rows
      constexpr int rs = decltype(RS)::value;
      const int wv = (int)((uint32_t)((wave * 64 + lane) * SVEC) | pend_wbad);
      const int ws = __builtin_amdgcn_readfirstlane((int)pend_woff);
#pragma unroll
      for (int i = 0; i < SW; ++i) {
        if constexpr (SVEC == 16)
          wreg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wv + i * (NW * 64 * SVEC), ws, 0));
        else
          wreg[i] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(w_rsrc, wv + i * (NW * 64 * SVEC), ws, 0));
      }
      const uint32_t aoff = (uint32_t)pend_idx * (uint32_t)(CIN * ES) + pend_acb + (uint32_t)(g4 * 16);
#pragma unroll
      for (int i = 0; i < ALD; ++i)
        aring[rs][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)(aoff + 64 * i), 0, 0));
      has[rs] = pend_has;
    };
    auto stage_write = [&](int par) {                    // wreg -> slab[par]
      char* dst = smem + par * SLAB + (wave * 64 + lane) * SVEC;
#pragma unroll
      for (int i = 0; i < SW; ++i) *reinterpret_cast<svec_t*>(dst + i * (NW * 64 * SVEC)) = wreg[i];
    };

    f32x4 acc[NS][2];
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) acc[sidx][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto compute = [&](auto RS, int par) {
      constexpr int rs = decltype(RS)::value;
      if (has[rs]) {
        const char* sl = smem + par * SLAB + lane * 16;
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) {
          f32x4 bfr[BLD];
#pragma unroll
          for (int i = 0; i < BLD; ++i) bfr[i] = *reinterpret_cast<const f32x4*>(sl + sidx * ITEM_BYTES + i * 1024);
          if constexpr (BF16) {
            const bf16x8_t av = __builtin_bit_cast(bf16x8_t, aring[rs][0]);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
              acc[sidx][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bfr[nt]), av, acc[sidx][nt], 0, 0, 0);
          } else {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
              for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                  acc[sidx][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[nt * 2 + tt][u], aring[rs][tt][u], acc[sidx][nt], 0, 0, 0);
          }
        }
      }
    };

    // ---- prologue: W(0) in slab 0, W(1) in registers, A(0), A(1) in flight, item 2 pending
    generate();
    issue(std::integral_constant<int, 0>{});
    generate();
    stage_write(0);
    issue(std::integral_constant<int, 1>{});
    generate();
    __syncthreads();
    int par = 0;
    for (int it = 0; it < n_steps; ++it) {
      [&]<int... Is>(std::integer_sequence<int, Is...>) {
        ((stage_write(par ^ 1), issue(std::integral_constant<int, (Is + D - 1) % D>{}), generate(),
          __builtin_amdgcn_sched_barrier(0), compute(std::integral_constant<int, Is>{}, par),
          __builtin_amdgcn_sched_barrier(0), __syncthreads(), par ^= 1), ...);
      }(std::make_integer_sequence<int, D>{});
    }

    // ---- epilogue: BN scale/shift (+ReLU), one store per tile; optional per-group column sums
    const int32_t row = (own >> 31) ? p.perm[(int64_t)g * 16 + l15] : -1;
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx) {
      float sums[2][4];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int c0 = sidx * 32 + nt * 16 + 4 * g4;
        f32x4 v = acc[sidx][nt];
        if (p.scale) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + c0);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + c0);
          v = v * sc + sh;
        }
        if (p.relu) {
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = fmaxf(v[u], 0.f);
        }
        if (row >= 0) {
          if constexpr (BF16) {
            uint2 o;
            o.x = f2bf(v[0]) | (f2bf(v[1]) << 16);
            o.y = f2bf(v[2]) | (f2bf(v[3]) << 16);
            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + (int64_t)row * COUT + c0) = o;
          } else {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (int64_t)row * COUT + c0) = v;
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) sums[nt][u] = row >= 0 ? v[u] : 0.f;
      }
      if (p.psum && (own >> 31)) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            float sv = sums[nt][u];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sv += __shfl_xor(sv, o, 64);
            sums[nt][u] = sv;
          }
        if (l15 == 0) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            *reinterpret_cast<f32x4*>(p.psum + (int64_t)g * COUT + sidx * 32 + nt * 16 + 4 * g4) =
                (f32x4){sums[nt][0], sums[nt][1], sums[nt][2], sums[nt][3]};
        }
      }
    }
  }
}

template <int CIN, int COUT, bool BF16>
static int launch_wg(const SconvArgs& a, int64_t groups_hint, hipStream_t stream) {
  constexpr int NW = 4;
  constexpr int ES = BF16 ? 2 : 4;
  const size_t lds = 2 * (size_t)(32 * COUT * ES) + NW * 28 * 16 * sizeof(int32_t);
  static AttrOnce attr_done;                         // per instantiation; idempotent
  if (attr_done.need()) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sconv_wg_kernel<CIN, COUT, BF16, NW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done.mark(); 
  }
  const int64_t ntask = cdiv(groups_hint, NW);
  int64_t grid = std::min<int64_t>(std::max<int64_t>(ntask, 8), 16384);
  grid = (grid + 7) / 8 * 8;
  hipEvent_t* pev = prof_kernel_events();
  if (pev[0]) {      // bench.py roofline leg: time exactly this dispatch
    hipExtLaunchKernelGGL((sconv_wg_kernel<CIN, COUT, BF16, NW>), dim3((unsigned)grid), dim3(NW * 64), lds, stream, pev[0], pev[1], 0, a);
    pev[0] = pev[1] = nullptr;
  } else {
    hipLaunchKernelGGL((sconv_wg_kernel<CIN, COUT, BF16, NW>), dim3((unsigned)grid), dim3(NW * 64), lds, stream, a);
  }
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------ launcher
template <int CIN, int COUT, bool BF16, int KSP, int D, int G, bool SPLIT = false>
static int launch_rg_d(const SconvArgs& a, int64_t groups_hint, hipStream_t stream) {
  constexpr int NS = COUT / 32;
  constexpr int NW = KSP;                                // a workgroup = the waves that share a tile (see launch_dma_d)
  constexpr int NPIECE = (G * 27 * 4 + 63) / 64;
  const size_t lds = NW * (NPIECE * 256 + 16) * sizeof(int32_t) + (KSP > 1 ? 2 * NW * 2 * G * 64 * sizeof(f32x4) : 0);
  static AttrOnce attr_done;                         // per instantiation; idempotent
  if (attr_done.need() && lds > 48 * 1024) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sconv_rg_kernel<CIN, COUT, BF16, D, KSP, G, NW, SPLIT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done.mark(); 
  }
  const int64_t ntask = cdiv(cdiv(groups_hint, G) * NS * KSP, NW);
  // one workgroup per task up to a cap: the hardware dispatcher then balances the uneven tasks (a grid of only the
  // resident workgroups was 30 % slower: each loops over ~3 tasks and the slowest decides)
  int64_t grid = std::min<int64_t>(std::max<int64_t>(ntask, 8), 65536);
  grid = (grid + 7) / 8 * 8;
  hipEvent_t* pev = prof_kernel_events();
  if (pev[0]) {      // bench.py roofline leg: time exactly this dispatch
    hipExtLaunchKernelGGL((sconv_rg_kernel<CIN, COUT, BF16, D, KSP, G, NW, SPLIT>), dim3((unsigned)grid), dim3(NW * 64), lds, stream, pev[0], pev[1], 0, a);
    pev[0] = pev[1] = nullptr;
  } else {
    hipLaunchKernelGGL((sconv_rg_kernel<CIN, COUT, BF16, D, KSP, G, NW, SPLIT>), dim3((unsigned)grid), dim3(NW * 64), lds, stream, a);
  }
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}
// sel (tests / A-B measurements; every choice gives bitwise-identical results): 0 = product choice, 1 = register-ring
// kernel, 2 = register-ring kernel with two groups per wave, 3 = LDS-DMA kernel (fp32 maps; split-phase LDS fetch, 4 ring slots), 4 = LDS-DMA kernel with the fetch inside the item (3 slots), 9 = traced build
template <int CIN, int COUT, bool BF16>
static int launch_rg(const SconvArgs& a, int64_t groups_hint, hipStream_t stream, int sel, int level, int split) {
  constexpr int NS = COUT / 32, NCB = CIN / 32;
  // the input-channel blocks are always split over the waves of a workgroup (a function of the shape only, so results
  // never depend on launch sizes): measured faster at every level, 12 % on the 84 k-row 64->64 layer, 40 % on level 4.
  // Splitting the kernel offsets as well measured neutral to 50 % slower.
  constexpr int KSP = NCB >= 4 ? 4 : NCB;
  // prefetch depth (does not touch the arithmetic): few waves per SIMD => nothing else hides the gather latency, keep
  // 5 items in flight; a full chip prefers the smaller register footprint
  // levels >= 5 never fill the chip (batch 16: <= 240 groups): few waves per SIMD.  A function of the LAYER, not of a
  // capacity: eager plans, reserved (graph) plans, the per-layer table and rocprof all see the same kernel.  bf16 maps are the
  // batch-64 configuration (BASELINE configs[2]): level 5 still fills the chip there
  const bool small = level >= (BF16 ? 6 : 5);
  if constexpr (!BF16 && CIN == 128 && COUT == 128) {      // the tail levels' plan on split arithmetic (a.Wp = the split pack)
    // (ring depth 9 / 12 instead of 6: L6 k=3 21.7 / 22.5 vs 18.1 us, L6 k2s2 13.2 / 14.7 vs 8.8 — profiles/r06c_kw_sweep.txt)
    if (split) return launch_rg_d<CIN, COUT, false, KSP, 6, 1, true>(a, groups_hint, stream);
  }
  EGONN_REQUIRE(!split, EGONN_ERR_INVALID, "sconv: the per-tile kernel has split arithmetic for fp32 128->128 maps only");
  if constexpr (!BF16) {
    {
      if (sel == 3 || (sel == 0 && !small)) return launch_dma_d<CIN, COUT, KSP, 4, false, true>(a, groups_hint, stream);
      if (sel == 4) return launch_dma_d<CIN, COUT, KSP, 3>(a, groups_hint, stream);
      if constexpr ((CIN == 32 && COUT == 32) || (CIN == 64 && COUT == 64)) {
        if (sel == 9) return launch_dma_d<CIN, COUT, KSP, 4, true, true>(a, groups_hint, stream);
      }
    }
  }
  // bf16 maps, big launches: two groups per wave share the W fragments (the items are load-bound: W is two thirds of the
  // bytes an item moves) — 9-12 % on the 32-channel layers at batch 64; fp32 items are MFMA-bound and gain nothing
  if constexpr (BF16) {      // very large launches (batch 64, levels 1-2): four groups per wave, another 4 %
    if (sel == 0 && groups_hint * NS * KSP >= 32768) return launch_rg_d<CIN, COUT, BF16, KSP, 3, 4>(a, groups_hint, stream);
  }
  if ((sel == 2 || (BF16 && sel == 0)) && !small) return launch_rg_d<CIN, COUT, BF16, KSP, BF16 ? 4 : 3, 2>(a, groups_hint, stream);
  if (small) return launch_rg_d<CIN, COUT, BF16, KSP, 6, 1>(a, groups_hint, stream);
  return launch_rg_d<CIN, COUT, BF16, KSP, BF16 ? 4 : 3, 1>(a, groups_hint, stream);
}

unsigned long long* g_sconv_trace = nullptr;              // measurement hook (egonn_debug_set_trace)

bool sconv_rg_supported(int cin, int cout) {
  auto ok = [](int c) { return c == 32 || c == 64 || c == 128 || c == 256; };
  return ok(cin) && ok(cout);
}

// in: [n_in][cin]; rg: row-group tables of the map; Wp: pack_rg_weights(bf16 matching); groups_hint: host upper bound of
// the groups in use (sizes the persistent grid only; the kernel reads the true count from rg.meta[0]).
int sconv_rg_forward(const void* in, int64_t n_in_cap, const RowGroups& rg, int64_t groups_hint, const void* Wp, int cin,
                     int cout, int bf16, const float* scale, const float* shift, int relu, void* out, float* psum,
                     hipStream_t stream, int variant, int level, int split, int32_t* flags, const float* residual) {
  EGONN_REQUIRE(rg.built, EGONN_ERR_STATE, "sconv: row-group tables not built");
  EGONN_REQUIRE(sconv_rg_supported(cin, cout), EGONN_ERR_INVALID, "sconv: channel plan %d->%d not supported (32/64/128/256)", cin, cout);
  const uint64_t ib = (uint64_t)n_in_cap * cin * (bf16 ? 2 : 4);
  EGONN_REQUIRE(ib < (1ull << 32) - (1ull << 20), EGONN_ERR_INVALID,
                "sconv: input feature map of %lld rows exceeds the 4 GiB buffer-resource range", (long long)n_in_cap);
  if (groups_hint <= 0) return EGONN_OK;
  SconvArgs a;
  a.in = in; a.snbr = rg.snbr; a.gmask = rg.gmask; a.perm = rg.perm; a.meta = rg.meta; a.Wp = Wp;
  a.scale = scale; a.shift = shift; a.out = out; a.psum = psum;
  a.in_bytes = (uint32_t)ib;
  a.w_bytes = (uint32_t)((uint64_t)rg.K * cin * cout * (bf16 ? 2 : 4));
  a.K = rg.K; a.relu = relu ? 1 : 0; a.cap_groups = rg.cap_groups;
  static const bool no_order = getenv("EGONN_NO_TASK_ORDER") != nullptr;   // (measurement switch)
  a.order4 = no_order ? nullptr : rg.order4;
  a.trace = variant == 9 ? g_sconv_trace : nullptr;
  a.flags = flags;
  a.res = residual;
  EGONN_REQUIRE(!residual || (!bf16 && level >= 5 && (variant == 0 || variant == 1)), EGONN_ERR_INVALID,
                "sconv: the epilogue residual exists in the per-tile kernel of fp32 maps (levels >= 5) and in the split kernel");

  // Measured (profiles/r02b_sconv.json, batch 16): in fp32 the per-wave kernel wins everywhere (the lock-step of the
  // cooperative kernel costs more than its saved W traffic when an item is 16-64 MFMAs of 32 cycles); with bf16 maps the
  // items are load-bound and the cooperative kernel wins on the big layers with >= 64 input or output channels.
  const int gsel = variant == 1 ? 1 : (variant == 4 ? 2 : (variant == 5 ? 3 : (variant == 6 ? 4 : (variant == 9 ? 9 : 0))));
  const bool coop = variant == 2 || (variant == 0 && bf16 && level <= 4 && cin * cout >= 32 * 64);   // (a function of the layer)
#define EGONN_RG_CASE(CI, CO)                                                                      \
  if (cin == CI && cout == CO) {                                                                   \
    if (coop) return bf16 ? launch_wg<CI, CO, true>(a, groups_hint, stream) : launch_wg<CI, CO, false>(a, groups_hint, stream); \
    return bf16 ? launch_rg<CI, CO, true>(a, groups_hint, stream, gsel, level, split) : launch_rg<CI, CO, false>(a, groups_hint, stream, gsel, level, split); \
  }
  EGONN_RG_CASE(32, 32)
  EGONN_RG_CASE(32, 64)
  EGONN_RG_CASE(64, 64)
  EGONN_RG_CASE(64, 128)
  EGONN_RG_CASE(128, 128)
  EGONN_RG_CASE(64, 32)       // input gradients of the 32->64 / 64->128 layers (training)
  EGONN_RG_CASE(128, 64)
  EGONN_RG_CASE(256, 256)     // MinkLoc3D lateral / transposed convolutions (models/minkfpn.py:50-52)
  EGONN_RG_CASE(128, 256)
  EGONN_RG_CASE(256, 128)
  EGONN_RG_CASE(64, 256)
  EGONN_RG_CASE(256, 64)
  EGONN_RG_CASE(32, 128)
  EGONN_RG_CASE(128, 32)
  EGONN_RG_CASE(32, 256)
  EGONN_RG_CASE(256, 32)
#undef EGONN_RG_CASE
  set_error("sconv: channel plan %d->%d has no instantiation", cin, cout);
  return EGONN_ERR_INVALID;
}

// Name of the kernel sconv_map dispatches for this layer (the profiler tags carry it, so that bench.py's dominant kernel is
// the kernel rocprofv3 names).  Mirrors the choices in sconv_map / sconv_rg_forward / launch_rg.
const char* sconv_kernel_name(const Ctx* ctx, int kind, int level, int cin, int cout, int bf16) {
  const int variant = ctx->conv_variant;
  if (sconv_uses_split(cin, cout, bf16, level, variant, ctx->split_max_level, kind)) return "sconv_split_kernel";
  const bool small = level >= (bf16 ? 6 : 5);
  const bool coop = variant == 2 || (variant == 0 && bf16 && level <= 4 && cin * cout >= 32 * 64);
  if (coop) return "sconv_wg_kernel";
  if (!bf16 && (variant == 5 || variant == 6 || variant == 9 || (variant == 0 && !small))) return "sconv_dma_kernel";
  return "sconv_rg_kernel";
}

// Arithmetic of an fp32 sparse convolution (ctx->conv_variant; egonn_debug_set_naive_conv):
//   0           product choice: split-bf16 kernel (sconv_split.hip) on the maps of levels <= ctx->split_max_level where it is
//               instantiated, exact fp32 MFMA kernels elsewhere
//   1..9        the exact fp32 kernels of this file (3 = plain one-thread-per-output kernel)
//   1000 + cfg  split-bf16 kernel with an explicit configuration (cfg 0 = its default)
// The product choice is a function of the LAYER (output level, channel plan), never of the batch or of a capacity: a scan
// gives bitwise the same descriptors alone, in any batch, and under eager or reserved (hipGraph) plans.
// Measured (profiles/r03e_split_threshold.txt, batch 16, four batches in flight): one launch alone, the lock-step split
// kernel wins from ~4 000 row groups up (L1 k3 60 -> 54 us, L2 64->64 89 -> 79) and loses below; with batches in flight it
// pays much earlier because it leaves the matrix pipe to the other batches: scans/s with the split kernel on launches of
// >= inf / 4096 / 2000 / 700 / 200 groups = 21.8 k / 23.3 k / 24.3 k / 24.8 k / 23.6 k, i.e. levels <= 0 / 2 / 3 / 4 / 6.
static int split_level_limit(int split_max_level) {      // the context's limit, or EGONN_SPLIT_MAX_LEVEL (measurement override)
  static const int env_level = [] {
    const char* e = getenv("EGONN_SPLIT_MAX_LEVEL");
    return e ? atoi(e) : -1;
  }();
  return (env_level >= 0 && split_max_level >= 0) ? env_level : split_max_level;
}
bool sconv_uses_split(int cin, int cout, int bf16, int level, int variant, int split_max_level, int kind) {
  if (bf16 || !sconv_split_supported(cin, cout)) return false;
  if (variant >= 1000) return true;
  if (variant != 0) return false;
  static const int env_k8 = [] {                          // EGONN_SPLIT_MAX_LEVEL_K8: measurement override for the 8-slot maps
    const char* e = getenv("EGONN_SPLIT_MAX_LEVEL_K8");
    return e ? atoi(e) : -1;
  }();
  if (kind != 0 && env_k8 >= 0) return level <= env_k8;
  return level <= split_level_limit(split_max_level);
}

// Offset-split rule (see kernels.h): a function of (map kind, output level) only.
//   kw      offset parts INSIDE a workgroup (sconv_split_kernel<..., KW>): the small maps are chains of K * Cin/32 dependent steps on
//           workgroups of one wave per SIMD; KW waves per SIMD walk KW interleaved shares of the offsets and hide each other's LDS
//           and MFMA latencies.  Measured per layer, batch 16 (profiles/r06c_kw_sweep.txt): KW 1 / 2 / 3 / 4: L4 128->128 70 / 50 / 42 /
//           40 us, L5 k=3 67 / 40 / 35 / 32, L5 k=2,s=2 27 / 19 / 17 / 16, L3 64->64 42 -> 27 (KW 2, 64 columns per workgroup).
//           The rule is KW = 2 on levels 3-5, both map classes: a KW 3-4 workgroup (12-16 waves, 120-160 KB of LDS) needs an empty
//           CU, and with four batches in flight the step pays for it — one box, scans/s / one-batch latency / aggregate: no parts
//           27.9 k / 1.196 ms / 0.209, KW 2 27.1 k / 1.060 / 0.263, KW (2,3,4) 25.7 k / 1.048 / 0.276.  Levels 6-7 stay on the
//           per-tile kernel (KW 3 on the lock-step kernel: 34 vs 23 us).
//   kparts  offset parts as SEPARATE workgroups + a reducer launch (blockIdx.z): built, parity-green, measured
//           (L5 k=3 65 -> 30 us, L4 128->128 67 -> 48), default off: the partial tiles are 16 MB per level-5 convolution and the step
//           with four batches in flight loses 4-6 % against 1.5-3 % for the in-workgroup parts at the same serial gain.
// EGONN_KSPLIT / EGONN_KSPLIT8 (kparts of the k=3 / 8-slot maps), EGONN_KSPLIT_KW / EGONN_KSPLIT_KW8, EGONN_KSPLIT_PARTS (column
// parts per task, 0 = automatic): measurement overrides, comma lists indexed by the output level.
void sconv_ksplit_defaults(KsRule* r) {
  static const KsRule rule = [] {
    KsRule q = {{{1, 1, 1, 1, 1, 1, 1, 1}, {1, 1, 1, 1, 1, 1, 1, 1}}, {{0, 0, 0, 2, 2, 2, 0, 0}, {0, 0, 0, 2, 2, 2, 0, 0}}, {0, 0, 0, 0, 0, 0, 0, 0}};
    auto parse = [](const char* name, int8_t* dst) {
      const char* e = getenv(name);
      for (int l = 0; e && *e && l < EGONN_NUM_LEVELS; ++l) {
        dst[l] = (int8_t)atoi(e);
        e = strchr(e, ',');
        if (e) ++e;
      }
    };
    parse("EGONN_KSPLIT", q.kparts[0]);
    parse("EGONN_KSPLIT8", q.kparts[1]);
    parse("EGONN_KSPLIT_KW", q.kw[0]);
    parse("EGONN_KSPLIT_KW8", q.kw[1]);
    parse("EGONN_KSPLIT_PARTS", q.col_parts);
    return q;
  }();
  *r = rule;
}
void sconv_ksplit_rule(const Ctx* ctx, int kind, int level, int* kparts, int* col_parts, int* kw) {
  const KsRule& rule = ctx->ks_rule;
  const int l = std::min(std::max(level, 0), EGONN_NUM_LEVELS - 1);
  const int K = kind == 0 ? 27 : 8, mc = kind == 0 ? 0 : 1;
  *kparts = std::min(std::max((int)rule.kparts[mc][l], 1), K);
  *col_parts = rule.col_parts[l];
  if (kw) *kw = rule.kw[mc][l];
}
size_t sconv_ksplit_scratch_floats(const Ctx* ctx) {
  size_t need = 0;
  for (int kind = 0; kind <= 2; ++kind)
    for (int l = (kind == 2 ? 0 : 1); l < EGONN_NUM_LEVELS - (kind == 2 ? 1 : 0); ++l) {
      int kp, cp;
      sconv_ksplit_rule(ctx, kind, l, &kp, &cp);
      if (kp > 1) need = std::max(need, (size_t)kp * rowgroup_cap_groups(ctx->plan, l) * 16 * 128);
    }
  return need;
}

int sconv_map(Ctx* ctx, int kind, int level, const void* in, const float* W, const void* Wp, const void* Wsp, int cin, int cout,
              int bf16, const float* scale, const float* shift, int relu, void* out, float* psum, float* scratch,
              size_t scratch_floats, hipStream_t stream) {
  Plan& P = ctx->plan;
  EGONN_REQUIRE(kind >= 0 && kind <= 2, EGONN_ERR_INVALID, "sconv: map kind %d", kind);
  const int lin = kind == 0 ? level : (kind == 1 ? level - 1 : level + 1);
  EGONN_REQUIRE(level >= 0 && level < EGONN_NUM_LEVELS && lin >= 0 && lin < EGONN_NUM_LEVELS, EGONN_ERR_INVALID,
                "sconv: level %d out of range for map kind %d", level, kind);
  Level& V = P.lv[level];
  if (V.n == 0) return EGONN_OK;
  const int K = kind == 0 ? 27 : 8;
  if (ctx->conv_variant == 3 || !sconv_rg_supported(cin, cout)) {
    EGONN_REQUIRE(!bf16, EGONN_ERR_INVALID, "sconv: bf16 feature maps need a 32/64/128/256-channel plan (%d->%d)", cin, cout);
    EGONN_REQUIRE(W, EGONN_ERR_INVALID, "sconv: the plain kernel needs the reference-layout kernel");
    if (kind == 2 && level == 0) EGONN_TRY(ensure_level0_parent_table(ctx, stream));
    const int32_t* nbr = kind == 0 ? V.nbr27 : (kind == 1 ? V.nbr8 : V.nbrT);
    EGONN_REQUIRE(!psum && !ctx->conv_residual, EGONN_ERR_INVALID, "sconv: group sums / epilogue residuals are produced by the MFMA kernels only");
    return sconv_naive(reinterpret_cast<const float*>(in), nbr, W, scale, shift, relu, reinterpret_cast<float*>(out), V.n, K,
                       cin, cout, stream);
  }
  EGONN_TRY(ensure_rowgroups(ctx, &kind, &level, 1, stream));
  const RowGroups& rg = kind == 0 ? V.rg27 : (kind == 1 ? V.rg8 : V.rgT);
  if (sconv_uses_split(cin, cout, bf16, level, ctx->conv_variant, ctx->split_max_level, kind)) {
    if (!Wsp) {   // stand-alone operator call: pack into the caller's scratch
      const size_t wn = (split_weights_bytes(K, cin, cout) + 3) / 4;
      EGONN_REQUIRE(W && scratch && scratch_floats >= wn, EGONN_ERR_STATE, "sconv: no scratch to pack the kernel into");
      EGONN_TRY(pack_split_weights(W, K, cin, cout, 0, 0, scratch, stream));
      Wsp = scratch;
    }
    int kparts = 1, col_parts = 0, kw = 0;
    sconv_ksplit_rule(ctx, kind, level, &kparts, &col_parts, &kw);
    if (ctx->gated_in2) { kparts = 1; kw = 0; }
    if (cin < 64) kw = 0;
    return sconv_split_forward(reinterpret_cast<const float*>(in), P.cap[lin], rg, rg.cap_groups, Wsp, cin, cout, scale, shift,
                               relu, reinterpret_cast<float*>(out), psum, stream,
                               ctx->conv_variant >= 1000 ? ctx->conv_variant - 1000 : 0, ctx->split_io, ctx->gated_in2, ctx->gated_gate,
                               P.batch, kparts, ctx->ks_part, ctx->ks_part_floats, col_parts, kw, ctx->dev_flags,
                               ctx->operand_autoscale ? reinterpret_cast<uint32_t*>(ctx->dev_counts + 24) : nullptr,
                               ctx->operand_autoscale ? P.lv[lin].n * cin : 0, ctx->conv_residual);
  }
  EGONN_REQUIRE(ctx->split_io == 0 && !ctx->gated_in2, EGONN_ERR_STATE,
                "sconv: split-form maps and gated inputs are read and written by the split kernel only");
  // fp32 maps of the tail levels (above split_max_level): the per-tile kernel on split arithmetic (a function of the layer)
  static const bool tail_split_ok = getenv("EGONN_NO_TAIL_SPLIT") == nullptr;          // measurement switch
  if (tail_split_ok && !bf16 && cin == 128 && cout == 128 && ctx->conv_variant == 0 && ctx->split_max_level >= 0 &&
      level > split_level_limit(ctx->split_max_level) && !ctx->operand_autoscale) {
    if (!Wsp) {
      const size_t wn = (split_weights_bytes(K, cin, cout) + 3) / 4;
      EGONN_REQUIRE(W && scratch && scratch_floats >= wn, EGONN_ERR_STATE, "sconv: no scratch to pack the kernel into");
      EGONN_TRY(pack_split_weights(W, K, cin, cout, 0, 0, scratch, stream));
      Wsp = scratch;
    }
    return sconv_rg_forward(in, P.cap[lin], rg, rg.cap_groups, Wsp, cin, cout, 0, scale, shift, relu, out, psum, stream, 0, level, 1,
                            ctx->dev_flags, ctx->conv_residual);
  }
  if (!Wp) {      // stand-alone operator call: pack into the caller's scratch
    const size_t wn = (size_t)K * cin * cout;
    EGONN_REQUIRE(W && scratch && scratch_floats >= wn, EGONN_ERR_STATE, "sconv: no scratch to pack the kernel into");
    EGONN_TRY(pack_rg_weights(W, K, cin, cout, bf16, 0, 0, scratch, stream));
    Wp = scratch;
  }
  return sconv_rg_forward(in, P.cap[lin], rg, rg.cap_groups, Wp, cin, cout, bf16, scale, shift, relu, out, psum, stream,
                          ctx->conv_variant, level, 0, nullptr, ctx->conv_residual);
}

}  // namespace egonn
