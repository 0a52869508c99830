// fp32 sparse convolution on the fp16 matrix pipe: two-way split operands, three products, fp32 accumulate (gfx950).
//
// Replaces the same reference operators as sconv.hip (MinkowskiConvolution k=3 / k=2,s=2 and
// MinkowskiConvolutionTranspose forward: models/minkgl.py:39,100,105 and :46-60; ME BasicBlock conv1/conv2 via
// layers/eca_block.py:58-63) for fp32 feature maps:
//
//   out[o] = act( (sum_k in[nbr[o][k]] @ W[k]) * scale + shift )
//
// Why.  v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate (157 TFLOP/s, 1/16 of the 16-bit matrix rate) and was the binding
// roof of every fp32 layer with >= 64 channels.  Round 3 ran these layers on v_mfma_f32_16x16x32_bf16 with
// both operands split EXACTLY into three bf16 parts and six products; the kernels then turned out to be bound by their own
// instruction stream (44 of 68 vector instructions per group and step were the split, profiles/r03c_pmc_wide.txt).  Round 5: fp16 parts.
// An fp16 number carries 11 significand bits, so  x = hi + lo + r  with hi = rn16(x), lo = rn16(x - hi), |r| <= 2^-22 |x|
// (fp32 itself rounds at 2^-24), an fp16 x fp16 product is exact in fp32, and
//     a * w = ah*wh + ah*wl + al*wh  (+ al*wl + cross terms with r: <= 3 * 2^-22 |a w|, dropped)
// is THREE products on v_mfma_f32_16x16x32_f16 (same rate as bf16) instead of six, a 24-instruction split instead of 44, and
// 4 bytes per weight instead of 6.  Range: fp16 holds |x| < 65504.  Weights are scaled by a power of two per kernel at pack
// time (max |W| -> [2^13, 2^14); exact, undone exactly in the epilogue) so that their low parts stay normal numbers;
// activations are used as they are: an fp32 activation beyond +-65504 becomes Inf in the split — every accumulator that gathers it
// then is Inf / NaN and the epilogue raises the plan's range flag (SplitArgs::flags; egonn_plan_status: EGONN_STATUS_FP16_RANGE;
// egonn_ctx_set_exact_fp32 selects the exact kernels of sconv.hip) — and the low part of an activation below 2^-3 is a subnormal
// with an ABSOLUTE error <= 2^-25: nothing next to the rounding of the sums these layers produce (operands far below 1 — input
// gradients — are scaled per launch: SplitArgs::in_maxbits).  Measured against the plain fp32 kernel
// (tests/test_gpu_graph.py::test_split_conv_matches_exact_fp32, tests/test_gpu_range.py): DESIGN.md §3.1.
//
// Decomposition (unchanged from round 3):
//   * a WORKGROUP of NW waves owns NW consecutive row groups (rowgroup.hip: 16 output rows each, sorted by neighbour mask
//     inside a window, so consecutive groups have nearly the same offsets present) and NSW 32-column slices;
//   * the workgroup walks the UNION of its groups' offsets k and the 32-channel blocks cb in lock-step.  Per step the slab
//     W[k][cb][its columns] (hi|lo fragments, 128*COUT bytes) is copied ONCE per workgroup global -> LDS by LDS-DMA
//     (buffer_load_dwordx4 ... lds, lane-linear = fragment order) and every wave reads its B fragments from there;
//   * every wave gathers the rows of its own group by LDS-DMA in full 128-byte lines (structured buffer: vindex = the
//     neighbour row from the table, -1 = absent = out of range = zeros without traffic; XOR-swizzled source chunk so that
//     the lane-linear LDS image is conflict-free for the ds_read_b128 fragment reads), splits them ONCE in registers
//     and feeds the MFMAs of all its column slices;
//   * DMA of step i+1 is issued at the start of step i; a wave waits for its own pieces at the end of the step and one
//     s_barrier per step publishes the slab (and retires the slab that is overwritten next);
//   * a group that lacks the offset skips the step's split and MFMAs (wave-uniform branch).
// Every output row is produced by one wave, summed in ascending k, ascending channel block, fixed term order: results do not
// depend on the batch, the grouping of other rows or eager vs graph execution (bitwise reproducible, batch-invariant).
// The compiler does not know that an LDS-DMA write feeds a later ds_read (it would drain vmcnt(0) in front of any LDS read
// it can see), so every LDS read of the step loop is issued from asm with its own counted waits.
// (The six-product bf16 kernel, its wave-wide variant and the 3 / 9-product measurement builds: tools/exp/r03_sconv_split_bf16x6.hip.)
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <utility>

#include "common.h"
#include <hip/hip_ext.h>
#include "kernels.h"

namespace egonn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

// channel of the 32-channel block that lane group g (= lane >> 4) holds in element e of its 8-element MFMA operand: the two
// 16-byte chunks g and 4+g of the gathered 128-byte row (the conflict-free ds_read_b128 pattern of the swizzled image)
__host__ __device__ static inline int sp_chan(int g, int e) { return e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4); }

// ------------------------------------------------------------------ weight packing
// W[k][ci][co] -> Wsp[k][cb][ns][f][lane][e] (fp16), f = 2*part + nt, part: 0 hi, 1 lo
//   = part( s * W[k][32cb + sp_chan(lane>>4, e)][32ns + 16nt + (lane&15)] ),   s = 2^(13 - exponent(max |W|))
// so that the slab of a step (k, cb) is one contiguous block of COUT/32 * 4 KB and every 1 KB piece is one lane-linear
// MFMA A-operand fragment.  Behind the fragments (byte offset K*cin*cout*4): float 1/s, uint32 bits of max |W|.
// flip / transpose: as pack_rg_weights (input-gradient kernels, k=2 <-> transposed pairs).
__global__ void split_absmax_kernel(const float* __restrict__ W, int64_t n, uint32_t* __restrict__ trailer) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float m = 0.f;
  for (int64_t i = t; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = fabsf(W[i]);
    m = (v == v && v < INFINITY) ? fmaxf(m, v) : m;       // NaN / Inf weights do not pick the scale
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(trailer + 1, __float_as_uint(m));
}
__device__ static inline float split_scale_of(uint32_t maxbits) {     // power of two s with s * max in [2^13, 2^14)
  const int e = (int)(maxbits >> 23) - 127;                            // max = 1.m * 2^e (0 / subnormal: e = -127 -> clamped)
  const int se = min(max(13 - e, -100), 100);
  return __uint_as_float((uint32_t)(se + 127) << 23);
}
__global__ void pack_split_weights_kernel(const float* __restrict__ W, int K, int cin, int cout, int flip, int transpose,
                                          uint16_t* __restrict__ out, uint32_t* __restrict__ trailer) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per_k = (int64_t)cin * cout;
  const float sc = split_scale_of(trailer[1]);
  if (t == 0) reinterpret_cast<float*>(trailer)[0] = 1.f / sc;
  if (t >= K * per_k * 2) return;
  const int ncb = cin / 32, ns_n = cout / 32;
  int64_t r = t;
  const int e = (int)(r & 7); r >>= 3;
  const int lane = (int)(r & 63); r >>= 6;
  const int f = (int)(r & 3); r >>= 2;
  const int ns = (int)(r % ns_n); r /= ns_n;
  const int cb = (int)(r % ncb); r /= ncb;
  const int k = (int)r;
  const int part = f >> 1, nt = f & 1;
  const int ci = 32 * cb + sp_chan(lane >> 4, e);
  const int co = 32 * ns + 16 * nt + (lane & 15);
  const int ks = flip ? K - 1 - k : k;
  const float v = sc * (transpose ? W[(int64_t)ks * per_k + (int64_t)co * cin + ci] : W[(int64_t)ks * per_k + (int64_t)ci * cout + co]);
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  out[t] = __builtin_bit_cast(uint16_t, part == 0 ? hi : lo);
}

size_t split_weights_bytes(int K, int cin, int cout) { return (size_t)K * cin * cout * 4 + 16; }
size_t sconv_split_part_floats(const RowGroups& rg, int cout, int kparts) {
  return kparts > 1 ? (size_t)kparts * rg.cap_groups * 16 * cout : 0;
}

int pack_split_weights(const float* W, int K, int cin, int cout, int flip, int transpose, void* out, hipStream_t stream) {
  EGONN_REQUIRE(cin % 32 == 0 && cout % 32 == 0, EGONN_ERR_INVALID, "sconv: channel counts must be multiples of 32 (%d->%d)", cin, cout);
  const int64_t nw = (int64_t)K * cin * cout, n = nw * 2;
  uint32_t* trailer = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(out) + nw * 4);
  HIP_CHECK(hipMemsetAsync(trailer, 0, 16, stream));
  hipLaunchKernelGGL(split_absmax_kernel, dim3((unsigned)std::min<int64_t>(cdiv(nw, 256), 256)), dim3(256), 0, stream, W, nw, trailer);
  hipLaunchKernelGGL(pack_split_weights_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, W, K, cin, cout, flip,
                     transpose, reinterpret_cast<uint16_t*>(out), trailer);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------ the kernel
struct SplitArgs {
  const float* in;           // [n_in][CIN] fp32
  const int32_t* snbr;       // row-group tables
  const uint32_t* gmask;
  const int32_t* perm;
  const int32_t* meta;       // [0] = groups in use
  const int32_t* order;      // dispatch order of the 4-group tasks (rowgroup.hip; nullable = table order)
  const void* Wsp;           // pack_split_weights
  const float* scale;        // folded BatchNorm (nullable)
  const float* shift;
  float* out;                // [n_out][COUT]
  float* psum;               // [groups][COUT] column sums of the stored values (nullable)
  uint32_t in_rows, w_bytes;
  int K, relu, cap_groups;
  int in_split, out_split;   // the input / output map is in SPLIT FORM: per row and 32-channel block the 128 bytes hold the fp16 hi
                             // parts (chunk g = channels sp_chan(g, 0..7)) and lo parts (chunk 4 + g) the consumer's MFMA operands
                             // are made of, instead of 32 fp32 values.  Same size, same gather; the consumer's in-loop split
                             // (24 vector instructions per 16 x 32 fragment) moves into the producer's epilogue.  hi / lo are
                             // computed by the same split8h either way: results are bitwise those of fp32 maps.
  // GATED instantiation only: the input map does not exist — row r of it is relu(in[r] * gate[scan][c] + in2[r]) (the tail of the
  // ECA block below, layers/eca_block.py:66-73), evaluated on the gathered fragments with the expression of eca_apply_kernel
  const float* in2 = nullptr;
  const float* gate = nullptr;            // [B][CIN]
  int B = 0;
  unsigned long long* trace = nullptr;   // measurement builds only (tools/split_trace.py): 12 u64 per wave task
  // OFFSET-SPLIT launches (kp_n > 1; maps below one round of the chip): blockIdx.z = kp owns the contiguous offset range
  // [kp * K / kp_n, (kp + 1) * K / kp_n), walks only its share of the steps and leaves the RAW accumulator tiles of every group
  // that has an offset in the range in part[kp][group][slice][tile][lane] (the accumulator layout itself: 1 KB coalesced stores).
  // sconv_split_reduce_kernel sums the parts in ascending kp and applies this kernel's epilogue — the hand-over is a kernel
  // boundary, nothing is synchronised in here.
  float* part = nullptr;
  int kp_n = 1;
  // RANGE GUARD.  fp16 parts hold |x| < 65520: a finite fp32 activation beyond that becomes Inf in split8h, every accumulator
  // that gathers it becomes Inf or NaN (Inf * 0, Inf - Inf), and the epilogue — which sees every accumulator before BN / ReLU can
  // hide it — raises bit 3 of the plan's flag word (egonn_plan_status: EGONN_STATUS_FP16_RANGE; egonn_ctx_set_exact_fp32 selects
  // the exact kernels).  Non-finite INPUTS raise it too: the flag says "this launch's fp32 semantics are not guaranteed".
  int32_t* flags = nullptr;
  // OPERAND SCALE (the input-gradient convolutions of a training step: gradients of 1e-6 .. 1e-8 would lose their low parts): the bits
  // of max |in| (split_absmax_kernel); the gathered rows are multiplied by the power of two that puts the maximum into
  // [2^13, 2^14) before they are split — what pack_split_weights does to the kernel — and the epilogue divides it out (both exact).
  const uint32_t* in_maxbits = nullptr;
  // RESIDUAL (nullable): out[row] = epilogue value + res[row] ([n_out][COUT] fp32) — the FPN step of MinkHead, `tconv(y) + conv1x1(x)`
  // (models/minkgl.py:46-60), as the transposed convolution's epilogue
  const float* res = nullptr;
};
__host__ __device__ static inline uint32_t ks_range_mask(int kp, int kp_n, int K) {      // offsets [kp*K/kp_n, (kp+1)*K/kp_n)
  const int k0 = kp * K / kp_n, k1 = (kp + 1) * K / kp_n;
  return ((1u << k1) - 1u) & ~((1u << k0) - 1u);
}

__device__ static inline float sp_row16_sum(float v) {   // sum over the 16 lanes of a DPP row (= the 16 rows of a tile)
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

// fp32 x 8 -> (hi, lo) fp16 x 8, round to nearest even at both levels (v_cvt_pk_f16_f32): 24 VALU
__device__ static inline void split8h(const f32x4& a0, const f32x4& a1, f16x8_t& hi, f16x8_t& lo) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float x0 = p < 2 ? a0[2 * p] : a1[2 * p - 4], x1 = p < 2 ? a0[2 * p + 1] : a1[2 * p - 3];
    const f16x2_t h = __builtin_convertvector((f32x2){x0, x1}, f16x2_t);
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    const f16x2_t l = __builtin_convertvector((f32x2){x0 - hf[0], x1 - hf[1]}, f16x2_t);
    hi[2 * p] = h[0]; hi[2 * p + 1] = h[1];
    lo[2 * p] = l[0]; lo[2 * p + 1] = l[1];
  }
}

// The epilogue of one 16-row x 32-column accumulator pair (tiles nt = 0, 1 of column slice ns): undo the weight pack scale, BN
// scale/shift (+ReLU), one 16-byte store per tile (fp32 map) or the split-form chunks, optional per-group column sums.  Shared by
// the convolution kernel and by the reducer of the offset-split launches (same expression, same order: the reducer's output for
// kp_n = 1 partials would be bitwise the kernel's own).
template <int COUT>
__device__ static inline void split_epilogue(const SplitArgs& p, const f32x4& a0, const f32x4& a1, int ns, int32_t row, int gw,
                                             int l15, int g4) {
  float winv = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.Wsp) + p.w_bytes);   // 1 / (pack scale): a power of two
  if (p.in_maxbits) winv *= 1.f / split_scale_of(p.in_maxbits[0]);                                     // (uniform) 1 / (operand scale)
  float sums[2][4];
  f32x4 vv[2];
  {
    // range guard: (x - x) is 0 for finite x and NaN for Inf / NaN, and NaN is sticky under addition — one test for the eight values
    const f32x4 z = (a0 - a0) + (a1 - a1);
    const float zz = (z[0] + z[1]) + (z[2] + z[3]);
    if (__builtin_expect(zz != 0.f, 0) && p.flags) atomicOr(p.flags, 8);
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int c0 = ns * 32 + nt * 16 + 4 * g4;
    f32x4 v = (nt == 0 ? a0 : a1) * winv;
    if (p.scale) {
      const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + c0);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + c0);
      v = v * sc + sh;
    }
    if (p.relu) {
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = fmaxf(v[u], 0.f);
    }
    if (p.res && row >= 0) v += *reinterpret_cast<const f32x4*>(p.res + (int64_t)row * COUT + c0);
    if (row >= 0 && !p.out_split) *reinterpret_cast<f32x4*>(p.out + (int64_t)row * COUT + c0) = v;
    vv[nt] = v;
#pragma unroll
    for (int u = 0; u < 4; ++u) sums[nt][u] = row >= 0 ? v[u] : 0.f;
  }
  if (p.out_split && row >= 0) {
    // this lane holds exactly the eight channels sp_chan(g4, 0..7) of its row: columns 4 g4.. of tile 0 and of tile 1
    f16x8_t oh, ol;
    split8h(vv[0], vv[1], oh, ol);
    char* o = reinterpret_cast<char*>(p.out) + ((int64_t)row * COUT + ns * 32) * 4 + 16 * g4;
    *reinterpret_cast<f16x8_t*>(o) = oh;
    *reinterpret_cast<f16x8_t*>(o + 64) = ol;
  }
  if (p.psum) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int u = 0; u < 4; ++u) sums[nt][u] = sp_row16_sum(sums[nt][u]);
    if (l15 == 0) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        *reinterpret_cast<f32x4*>(p.psum + (int64_t)gw * COUT + ns * 32 + nt * 16 + 4 * g4) =
            (f32x4){sums[nt][0], sums[nt][1], sums[nt][2], sums[nt][3]};
    }
  }
}

// ------------------------------------------------------------------ reducer of the offset-split launches
// One wave per (group, 32-column slice): the parts of the group's tile pair that exist (the group has an offset in the part's
// range: the same test the convolution kernel stores under) are summed in ASCENDING kp — a fixed partition and a fixed order, so
// a row's result depends on nothing but its own neighbours — and finished by split_epilogue.  Loads of up to nine parts are in
// flight at once.
template <int COUT>
__global__ __launch_bounds__(COUT * 2) void sconv_split_reduce_kernel(const SplitArgs p) {
  constexpr int NSTOT = COUT / 32;
  const int lane = threadIdx.x & 63;
  const int ns = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, g4 = lane >> 4;
  const int gw = blockIdx.x;                                   // (< cap_groups: the grid)
  // header and parts requested together: group count, mask, output rows, and up to nine parts whether they exist or not (what a
  // part without the group's offsets holds is never used: selected by the mask below) — one round trip for a launch that is
  // nothing but round trips
  const __amdgpu_buffer_rsrc_t m_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(p.meta), 0, 4, 0x00020000);
  const int ng_v = __builtin_amdgcn_raw_buffer_load_b32(m_rsrc, 0, 0, 0);
  const __amdgpu_buffer_rsrc_t g_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(p.gmask), 0, p.cap_groups * 4, 0x00020000);
  const uint32_t gm_v = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(g_rsrc, 0, gw * 4, 0);
  const int32_t row_v = p.perm[(int64_t)gw * 16 + l15];
  const float* src = p.part + ((int64_t)gw * NSTOT + ns) * 512 + lane * 4;
  const int64_t kp_stride = (int64_t)p.cap_groups * NSTOT * 512;
  f32x4 v0[9], v1[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    v0[i] = v1[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (i < p.kp_n) {                                          // uniform
      v0[i] = *reinterpret_cast<const f32x4*>(src + i * kp_stride);
      v1[i] = *reinterpret_cast<const f32x4*>(src + i * kp_stride + 256);
    }
  }
  const int ngroups = __builtin_amdgcn_readfirstlane(min(ng_v, p.cap_groups));
  const uint32_t gm = __builtin_amdgcn_readfirstlane(gm_v);
  if (gw >= ngroups || !(gm >> 31)) return;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const bool has = i < p.kp_n && (gm & ks_range_mask(i, p.kp_n, p.K)) != 0;      // wave-uniform
    a0 += has ? v0[i] : zero;
    a1 += has ? v1[i] : zero;
  }
  for (int c0 = 9; c0 < p.kp_n; c0 += 9) {                     // more than nine parts: the rest in rounds of nine, existing ones only
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int kp = c0 + i;
      const bool has = kp < p.kp_n && (gm & ks_range_mask(kp, p.kp_n, p.K)) != 0;
      v0[i] = v1[i] = zero;
      if (has) {
        v0[i] = *reinterpret_cast<const f32x4*>(src + kp * kp_stride);
        v1[i] = *reinterpret_cast<const f32x4*>(src + kp * kp_stride + 256);
      }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) { a0 += v0[i]; a1 += v1[i]; }
  }
  split_epilogue<COUT>(p, a0, a1, ns, row_v, gw, l15, g4);
}

template <int NSW, int NW, bool GATED = false, int KW = 1>
struct SplitGeom {
  static constexpr int NS = NSW;                         // 32-column slices a workgroup owns
  static constexpr int SLAB = NS * 4096;                 // bytes of W[k][cb][its columns]: hi | lo fragments
  static constexpr int NPIECE = NS * 4;                  // 1 KB DMA pieces per slab
  static constexpr int WPP = (NPIECE + NW - 1) / NW;     // pieces a wave issues per step (at most)
  static constexpr int TBL_BYTES = 27 * 64;              // one group's table: 27 offsets x 16 slots
  static constexpr int WAVE_LDS = 2048 + 2 * 2048 + (GATED ? 2 * 2048 : 0);   // table (padded) + two ring slots of 16 x 128 B (+ two of the second operand)
  static constexpr int WAVES_AT = KW * 2 * SLAB;         // two slabs per offset part
  static constexpr int LDS_BYTES = WAVES_AT + NW * KW * WAVE_LDS;
};

// NSW: 32-column slices per workgroup (grid.y = COUT/32/NSW column parts).  Small maps (levels 3-4: a few hundred tasks, less
// than one round of the chip) are a chain of K*CIN/32 dependent steps per task; giving every column part its own workgroup
// shortens the step (fewer MFMAs, a smaller slab) and multiplies the workgroups in flight; the rows are then gathered once
// per part, which small maps can afford.  Columns are independent: the results are bitwise the same for every NSW.
// KS: the offset-split instantiation (kp_n > 1).  One task per workgroup, task = blockIdx.x (a launch of a small map is one round
// of the chip: neither the XCD slices nor the longest-first order matter), and NOTHING in front of the header loads: group
// count, masks, neighbour table and output rows are requested together — the chain in front of the first gather is two
// global round trips (header, rows) instead of five (count, order, masks, table, rows).
template <int CIN, int COUT, int NW, int NSW, bool TRACE = false, bool GATED = false, bool KS = false, int KW = 1>
__global__ __launch_bounds__(NW * KW * 64) void sconv_split_kernel(const SplitArgs p) {
  using GEO = SplitGeom<NSW, NW, GATED, KW>;
  static_assert(KW == 1 || (!GATED && !TRACE), "offset parts inside a workgroup: plain instantiations only");
  static_assert(!GATED || CIN == 32, "gated input: one channel block");
  constexpr int NS = NSW, NSTOT = COUT / 32, NCB = CIN / 32;
  static_assert(NSTOT % NSW == 0, "column parts");
  const int ns0 = blockIdx.y * NSW;                      // first column slice of this workgroup
  const float in_sc = p.in_maxbits ? split_scale_of(__builtin_amdgcn_readfirstlane((int)p.in_maxbits[0])) : 1.f;   // operand scale (a power of two)
  constexpr int SLAB = GEO::SLAB, NPIECE = GEO::NPIECE, WPP = GEO::WPP;
  static_assert(GEO::LDS_BYTES <= 160 * 1024, "LDS budget");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  typedef __attribute__((address_space(3))) char lds_char;
  const int lane = threadIdx.x & 63;
  const int wave_id = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // KW > 1: the workgroup's waves are KW offset parts x NW groups.  Part kw walks the offsets k with k % KW == kw of its group's
  // task in lock-step with the other parts (own slab pair, own ring), so that a SIMD holds KW waves that hide each other's LDS
  // and MFMA latencies; the parts' accumulators are summed through LDS in fixed order (part 0 + part 1 + ...) after the loop.
  const int wave = KW == 1 ? wave_id : wave_id % NW;     // group of the task
  const int kw = KW == 1 ? 0 : wave_id / NW;             // offset part
  const int l15 = lane & 15, g4 = lane >> 4;
  const int K = p.K;
  char* const wl = smem + GEO::WAVES_AT + wave_id * GEO::WAVE_LDS;
  int32_t* const tbl = reinterpret_cast<int32_t*>(wl);
  char* const ring = wl + 2048;
  const uint32_t smem_addr = (uint32_t)(uintptr_t)(lds_char*)smem;
  const uint32_t wl_addr = (uint32_t)(uintptr_t)(lds_char*)wl;
  // operand read addresses inside a gathered group image: chunks g4 and 4+g4 of row l15 (XOR swizzle on the chunk index)
  const uint32_t rd0 = wl_addr + 2048 + (uint32_t)(l15 * 128 + ((g4 ^ (l15 & 7)) * 16));
  const uint32_t rd1 = wl_addr + 2048 + (uint32_t)(l15 * 128 + (((4 + g4) ^ (l15 & 7)) * 16));
  const uint32_t tb0 = wl_addr + (uint32_t)((lane >> 3) * 4);          // table entry of this lane's DMA rows (L>>3, 8+(L>>3))
  const uint32_t wrd = smem_addr + (uint32_t)(kw * 2 * SLAB + lane * 16);   // fragment read address inside a slab piece (this part's slab pair)
  const int dma_chunk = ((lane & 7) ^ (lane >> 3)) * 16;
  const int w_lane = lane * 16;

  const __amdgpu_buffer_rsrc_t a_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), (short)(CIN * 4), (int)p.in_rows, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Wsp), 0, (int)p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t a2_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(GATED ? p.in2 : p.in), (short)(CIN * 4), (int)p.in_rows, 0x00020000);

  // (KS: the group count travels with the header's vector loads — a scalar load would be waited for in front of them)
  int ngroups = 0;
  if constexpr (!KS) ngroups = __builtin_amdgcn_readfirstlane(min(p.meta[0], p.cap_groups));
  const int ntask = (ngroups + NW - 1) / NW;
  // contiguous eighth of the tasks per XCD (block b runs on XCD b % 8): a Z-order slice of the map per L2
  const int xcd = blockIdx.x & 7, nper = gridDim.x >> 3;
  const int cpx = (ntask + 7) >> 3;

  for (int lt = KS ? 0 : (int)(blockIdx.x >> 3); lt < (KS ? 1 : cpx); lt += nper) {
    const int tslot = KS ? (int)blockIdx.x : xcd * cpx + lt;
    if (!KS && tslot >= ntask) continue;                 // workgroup-uniform
    const int task = (!KS && NW == 4 && p.order) ? __builtin_amdgcn_readfirstlane(p.order[tslot]) : tslot;   // longest tasks first
    const int g0 = task * NW;
    const int gw = g0 + wave;                            // this wave's group
    unsigned long long tr[12] = {};
    auto now = [] { return (unsigned long long)__builtin_amdgcn_s_memtime(); };
    if constexpr (TRACE) tr[0] = now();
    // ---- header: masks of the workgroup's groups, this wave's neighbour table (two 16-byte pieces per lane) and output rows,
    // requested together (indices clipped to the tables' capacity; what lies beyond the groups in use is masked out below)
    const int gwc = min(gw, p.cap_groups - 1);
    uint32_t mload = 0;
    if (lane < NW) mload = p.gmask[min(g0 + lane, p.cap_groups - 1)];
    int4 hv0 = make_int4(-1, -1, -1, -1), hv1 = make_int4(-1, -1, -1, -1);
    {
      const int n16 = K * 4;
      const int4* src = reinterpret_cast<const int4*>(p.snbr + (int64_t)gwc * K * 16);
      if (lane < n16) hv0 = src[lane];
      if (lane + 64 < n16) hv1 = src[lane + 64];
    }
    int32_t orow = p.perm[(int64_t)gwc * 16 + l15];
    if constexpr (KS) {
      const __amdgpu_buffer_rsrc_t m_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(p.meta), 0, 4, 0x00020000);
      ngroups = __builtin_amdgcn_readfirstlane(min(__builtin_amdgcn_raw_buffer_load_b32(m_rsrc, 0, 0, 0), p.cap_groups));
    }
    // ---- masks: union over the workgroup (identical in every wave), own group
    if (lane >= NW || g0 + lane >= ngroups) mload = 0;
    uint32_t U = 0;
#pragma unroll
    for (int q = 0; q < NW; ++q) U |= (uint32_t)__builtin_amdgcn_readlane((int)mload, q);      // (lanes 0..NW-1 hold the masks)
    if (!(U >> 31)) continue;                            // nothing but padding groups
    uint32_t gm = (uint32_t)__builtin_amdgcn_readlane((int)mload, wave);
    const bool live = (gm >> 31) != 0;
    if constexpr (KS) {                                  // offset-split launch: this workgroup's share of the offsets (uniform)
      const uint32_t rm = ks_range_mask((int)blockIdx.z, p.kp_n, K);
      U &= rm;
      gm &= rm;
      if (!U) continue;
    }
    const uint32_t gm_all = gm;                          // (the group's offsets in this launch's range, all parts)
    int n_steps_wg = 0;                                  // KW > 1: lock-steps of the workgroup = the longest part's
    if constexpr (KW > 1) {
      constexpr uint32_t base = KW == 2 ? 0x5555555u : (KW == 3 ? 0x1249249u : 0x1111111u);     // k % KW == 0
#pragma unroll
      for (int q = 0; q < KW; ++q) n_steps_wg = max(n_steps_wg, __popc(U & (base << q) & 0x07FFFFFFu) * NCB);
      U &= base << kw;
      gm &= base << kw;
    }
    (void)gm_all;
    f32x4 gq0 = {0.f, 0.f, 0.f, 0.f}, gq1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (GATED) {
      // scan of this wave's group (a group never straddles two scans): last b with meta[1 + b] <= gw; the gate of the lane's channels
      int lo = 0, hi = p.B;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (p.meta[1 + mid] <= gw) lo = mid; else hi = mid;
      }
      gq0 = *reinterpret_cast<const f32x4*>(p.gate + (int64_t)lo * CIN + 4 * g4);
      gq1 = *reinterpret_cast<const f32x4*>(p.gate + (int64_t)lo * CIN + 16 + 4 * g4);
    }

    // ---- the group's neighbour table -> wave-private LDS; output rows of the epilogue
    if (!live) orow = -1;
    if (live) {
      int4* dst = reinterpret_cast<int4*>(tbl);
      dst[lane] = hv0;
      if (lane + 64 < 27 * 4) dst[lane + 64] = hv1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    if constexpr (TRACE) tr[1] = now();
    // ---- step generator (scalar, identical in every wave): set bits of the union x channel blocks
    uint32_t mk = U & 0x07FFFFFFu;
    const int n_steps = KW == 1 ? __popc(mk) * NCB : n_steps_wg;
    int gen_k = 0, gen_cb = 0;
    auto gen = [&](int& k, int& cb) {                      // k = 27: past the end
      const bool need = (gen_cb == 0);
      const bool take = need && (mk != 0);
      const bool valid = !need || take;
      gen_k = take ? __builtin_ctz(mk | 0x80000000u) : gen_k;
      mk = take ? (mk & (mk - 1)) : mk;
      k = valid ? gen_k : 27;
      cb = gen_cb;
      gen_cb = (valid && gen_cb + 1 < NCB) ? gen_cb + 1 : 0;
    };
    // this wave's share of the step's requests: its pieces of the slab, the rows of its group if it has the offset.
    // Requests that would move nothing are NOT issued (a step costs a wave 0..WPP+2 vector-memory instructions): the
    // texture path is the busiest unit of these kernels (profiles/r03c_pmc_wide.txt) and an out-of-range piece costs it
    // as much as a real one.  Nothing is counted: the wave drains its queue (vmcnt(0)) before the step's barrier.
    auto issue = [&](int slot, int k, int cb, int32_t i0, int32_t i1) {
      if (k >= 27) return;
      const int woff = ((k * NCB + cb) * NSTOT + ns0) * 4096;
#pragma unroll
      for (int q = 0; q < WPP; ++q) {
        const int pc = wave + NW * q;
        if (NPIECE % NW == 0 || pc < NPIECE)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_char*)(smem + (kw * 2 + slot) * SLAB + pc * 1024), 16, w_lane, woff + pc * 1024, 0, 0);
      }
      if ((gm >> k) & 1u) {
        __builtin_amdgcn_struct_ptr_buffer_load_lds(a_rsrc, (lds_char*)(ring + slot * 2048), 16, i0, dma_chunk, cb * 128, 0, 0);
        __builtin_amdgcn_struct_ptr_buffer_load_lds(a_rsrc, (lds_char*)(ring + slot * 2048 + 1024), 16, i1, dma_chunk, cb * 128, 0, 0);
        if constexpr (GATED) {
          __builtin_amdgcn_struct_ptr_buffer_load_lds(a2_rsrc, (lds_char*)(ring + 4096 + slot * 2048), 16, i0, dma_chunk, cb * 128, 0, 0);
          __builtin_amdgcn_struct_ptr_buffer_load_lds(a2_rsrc, (lds_char*)(ring + 4096 + slot * 2048 + 1024), 16, i1, dma_chunk, cb * 128, 0, 0);
        }
      }
    };
    auto idx_read = [&](int k, int32_t& i0, int32_t& i1) {   // issued; awaited by the lgkmcnt(0) in front of the barrier
      if (k < 27 && ((gm >> k) & 1u)) {
        const uint32_t tb = tb0 + (uint32_t)(k * 64);
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:32" : "=&v"(i0), "=&v"(i1) : "v"(tb) : "memory");
      }
    };

    f32x4 acc[NS][2];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) acc[ns][0] = acc[ns][1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto wread = [&](uint32_t a, auto NSI, f32x4 (&w)[4]) {      // the four fragments of column slice NSI (issued): hi nt0|nt1, lo nt0|nt1
      constexpr int off = decltype(NSI)::value * 4096;
      asm volatile(
          "ds_read_b128 %0, %4 offset:%5\n\t"
          "ds_read_b128 %1, %4 offset:%5+1024\n\t"
          "ds_read_b128 %2, %4 offset:%5+2048\n\t"
          "ds_read_b128 %3, %4 offset:%5+3072"
          : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
          : "v"(a), "n"(off)
          : "memory");
    };
    auto wwait = [&](f32x4 (&w)[4]) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3])::"memory");
    };
    auto mfma = [](const f32x4& wf, const f16x8_t& af, f32x4& c) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, wf), af, c, 0, 0, 0);
    };
    // ---- one step of this wave's group: rows of ring slot x slab of the same slot
    auto compute = [&](int slot) {
      f32x4 ra0, ra1, rb0, rb1, w[4];
      {
        const uint32_t r0 = rd0 + (uint32_t)(slot * 2048), r1 = rd1 + (uint32_t)(slot * 2048);
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "=&v"(ra0), "=&v"(ra1) : "v"(r0), "v"(r1) : "memory");
        if constexpr (GATED) {
          const uint32_t q0 = r0 + 4096u, q1 = r1 + 4096u;
          asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "=&v"(rb0), "=&v"(rb1) : "v"(q0), "v"(q1) : "memory");
        }
      }
      const uint32_t wa = wrd + (uint32_t)(slot * SLAB);
      wread(wa, std::integral_constant<int, 0>{}, w);
      if constexpr (GATED) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra0), "+v"(ra1), "+v"(rb0), "+v"(rb1)::"memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra0), "+v"(ra1)::"memory");
      wwait(w);
      f16x8_t ah, al;
      if constexpr (GATED) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          ra0[u] = fmaxf(ra0[u] * gq0[u] + rb0[u], 0.f);        // eca_apply_kernel's expression
          ra1[u] = fmaxf(ra1[u] * gq1[u] + rb1[u], 0.f);
        }
        split8h(ra0, ra1, ah, al);
      } else if (p.in_split) { ah = __builtin_bit_cast(f16x8_t, ra0); al = __builtin_bit_cast(f16x8_t, ra1); }   // (uniform branch)
      else {
        if (in_sc != 1.f) { ra0 *= in_sc; ra1 *= in_sc; }                                                       // (uniform branch)
        split8h(ra0, ra1, ah, al);
      }
      [&]<int... NSI>(std::integer_sequence<int, NSI...>) {
        (([&] {
           f32x4 wn[4];
           if constexpr (NSI + 1 < NS) wread(wa, std::integral_constant<int, NSI + 1>{}, wn);
           __builtin_amdgcn_sched_barrier(0);
           // small terms first; w[2*part + nt]
           mfma(w[2], ah, acc[NSI][0]); mfma(w[3], ah, acc[NSI][1]);        // w lo * a hi
           mfma(w[0], al, acc[NSI][0]); mfma(w[1], al, acc[NSI][1]);        // w hi * a lo
           mfma(w[0], ah, acc[NSI][0]); mfma(w[1], ah, acc[NSI][1]);        // hi * hi
           __builtin_amdgcn_sched_barrier(0);
           if constexpr (NSI + 1 < NS) {
             wwait(wn);
#pragma unroll
             for (int i = 0; i < 4; ++i) w[i] = wn[i];
           }
         }()), ...);
      }(std::make_integer_sequence<int, NS>{});
    };

    // ---- prologue: step 0 in flight, rows of step 1 looked up
    int k0, cb0, k1, cb1;
    int32_t i0 = -1, i1 = -1;
    gen(k0, cb0);
    idx_read(k0, i0, i1);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(i0), "+v"(i1)::"memory");
    issue(0, k0, cb0, i0, i1);
    gen(k1, cb1);
    idx_read(k1, i0, i1);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(i0), "+v"(i1)::"memory");
    if constexpr (TRACE) tr[2] = now();
    __builtin_amdgcn_s_barrier();
    if constexpr (TRACE) tr[3] = now();
    // ---- main loop: requests of step i+1, arithmetic of step i, drain, barrier
    for (int i = 0; i < n_steps; ++i) {
      const int slot = i & 1;
      unsigned long long ta, tb, tc, td;
      if constexpr (TRACE) ta = now();
      issue(slot ^ 1, k1, cb1, i0, i1);
      int k2, cb2;
      gen(k2, cb2);
      idx_read(k2, i0, i1);                              // (the DMA above has read its index registers at issue)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (TRACE) tb = now();
      const bool has = k0 < 27 && ((gm >> k0) & 1u) != 0;
      if (has) compute(slot);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (TRACE) tc = now();
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(i0), "+v"(i1)::"memory");
      if constexpr (TRACE) td = now();
      __builtin_amdgcn_s_barrier();
      if constexpr (TRACE) {
        tr[4] += tb - ta; tr[5] += tc - tb; tr[6] += td - tc; tr[7] += now() - td; tr[8] += has ? 1 : 0;
      }
      k0 = k1; k1 = k2; cb1 = cb2;
    }
    if constexpr (TRACE) tr[9] = now();

    // ---- KW > 1: the parts' tiles through LDS (the slabs are retired: the loop ended on a barrier), summed in part order
    if constexpr (KW > 1) {
      f32x4* const red = reinterpret_cast<f32x4*>(smem);
      if (kw > 0) {
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) {
          red[(((kw - 1) * NW + wave) * NS * 2 + ns * 2) * 64 + lane] = acc[ns][0];
          red[(((kw - 1) * NW + wave) * NS * 2 + ns * 2 + 1) * 64 + lane] = acc[ns][1];
        }
      }
      __syncthreads();
      if (kw == 0) {
#pragma unroll
        for (int q = 1; q < KW; ++q)
#pragma unroll
          for (int ns = 0; ns < NS; ++ns) {
            acc[ns][0] += red[(((q - 1) * NW + wave) * NS * 2 + ns * 2) * 64 + lane];
            acc[ns][1] += red[(((q - 1) * NW + wave) * NS * 2 + ns * 2 + 1) * 64 + lane];
          }
      }
    }
    // ---- epilogue: BN scale/shift (+ReLU), one 16-byte store per tile; optional per-group column sums
    if (live && kw == 0) {
      if constexpr (KS) {
        // offset-split launch: the raw tiles, in accumulator layout (sconv_split_reduce_kernel finishes them)
        if (gm_all) {
          float* dst = p.part + (((int64_t)blockIdx.z * p.cap_groups + gw) * NSTOT + ns0) * 512 + lane * 4;
#pragma unroll
          for (int ns = 0; ns < NS; ++ns) {
            *reinterpret_cast<f32x4*>(dst + ns * 512) = acc[ns][0];
            *reinterpret_cast<f32x4*>(dst + ns * 512 + 256) = acc[ns][1];
          }
        }
      } else {
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) split_epilogue<COUT>(p, acc[ns][0], acc[ns][1], ns0 + ns, orow, gw, l15, g4);
      }
    }
    if constexpr (TRACE) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      tr[10] = now();
      tr[11] = (unsigned long long)n_steps;
      if (lane == 0 && p.trace) {
        unsigned long long* o = p.trace + (int64_t)gw * 12;
#pragma unroll
        for (int q = 0; q < 12; ++q) o[q] = tr[q];
      }
    }
    __builtin_amdgcn_s_barrier();                        // the next task rewrites tables and slabs
  }
}

template <int COUT>
static int launch_split_reduce(const SplitArgs& a, int64_t groups_hint, hipStream_t stream, hipEvent_t ev_stop) {
  const dim3 grid((unsigned)groups_hint);
  if (ev_stop) hipExtLaunchKernelGGL((sconv_split_reduce_kernel<COUT>), grid, dim3(COUT * 2), 0, stream, nullptr, ev_stop, 0, a);
  else hipLaunchKernelGGL((sconv_split_reduce_kernel<COUT>), grid, dim3(COUT * 2), 0, stream, a);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

template <int CIN, int COUT, int NW, int NSW, bool TRACE = false, bool GATED = false, bool KS = false, int KW = 1>
static int launch_split(const SplitArgs& a, int64_t groups_hint, hipStream_t stream) {
  using GEO = SplitGeom<NSW, NW, GATED, KW>;
  static AttrOnce attr_done;
  if (attr_done.need()) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sconv_split_kernel<CIN, COUT, NW, NSW, TRACE, GATED, KS, KW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done.mark();
  }
  const int64_t ntask = cdiv(groups_hint, NW);
  int64_t gridx = std::min<int64_t>(std::max<int64_t>(ntask, 8), 1 << 20);
  gridx = (gridx + 7) / 8 * 8;
  const dim3 grid((unsigned)gridx, (unsigned)(COUT / 32 / NSW), (unsigned)a.kp_n);
  hipEvent_t* pev = prof_kernel_events();
  hipEvent_t ev_stop = nullptr;
  if (pev[0]) {      // bench.py roofline leg: time exactly this dispatch (an offset-split layer: begin of this one .. end of its reducer)
    ev_stop = pev[1];
    hipExtLaunchKernelGGL((sconv_split_kernel<CIN, COUT, NW, NSW, TRACE, GATED, KS, KW>), grid, dim3(NW * KW * 64), GEO::LDS_BYTES, stream,
                          pev[0], a.kp_n > 1 ? nullptr : pev[1], 0, a);
    pev[0] = pev[1] = nullptr;
  } else {
    hipLaunchKernelGGL((sconv_split_kernel<CIN, COUT, NW, NSW, TRACE, GATED, KS, KW>), grid, dim3(NW * KW * 64), GEO::LDS_BYTES, stream, a);
  }
  HIP_CHECK(hipGetLastError());
  if (a.kp_n > 1) return launch_split_reduce<COUT>(a, groups_hint, stream, ev_stop);
  return EGONN_OK;
}

// exactly the channel plans EGONN_SP_PLAN instantiates below (the forward plans of the trunk and heads + the two
// input-gradient plans); every other pair stays on the exact kernels of sconv.hip
bool sconv_split_supported(int cin, int cout) {
  static const int plans[][2] = {{32, 32}, {32, 64}, {64, 64}, {64, 128}, {128, 128}, {64, 32}, {128, 64}};
  for (const auto& p : plans)
    if (p[0] == cin && p[1] == cout) return true;
  return false;
}

// cfg = 100 + NW * 10 + 2 [+ 400 * (1 + log2(column parts))]; 0 = product choice (142: workgroups of 4 waves, automatic parts)
int sconv_split_forward(const float* in, int64_t n_in_cap, const RowGroups& rg, int64_t groups_hint, const void* Wsp, int cin,
                        int cout, const float* scale, const float* shift, int relu, float* out, float* psum, hipStream_t stream,
                        int cfg, int split_io, const float* gated_in2, const float* gated_gate, int B, int kparts, float* part,
                        size_t part_floats, int col_parts, int kw, int32_t* flags, uint32_t* in_absmax, int64_t in_elems, const float* residual) {
  EGONN_REQUIRE(rg.built, EGONN_ERR_STATE, "sconv: row-group tables not built");
  EGONN_REQUIRE(sconv_split_supported(cin, cout), EGONN_ERR_INVALID, "sconv(split): channel plan %d->%d not supported", cin, cout);
  EGONN_REQUIRE((uint64_t)n_in_cap * cin * 4 < (1ull << 32) - (1ull << 20), EGONN_ERR_INVALID,
                "sconv: input feature map of %lld rows exceeds the 4 GiB buffer-resource range", (long long)n_in_cap);
  if (groups_hint <= 0) return EGONN_OK;
  SplitArgs a;
  a.in = in; a.snbr = rg.snbr; a.gmask = rg.gmask; a.perm = rg.perm; a.meta = rg.meta; a.Wsp = Wsp;
  static const bool no_order = getenv("EGONN_NO_TASK_ORDER") != nullptr;   // (measurement switch)
  a.order = no_order ? nullptr : rg.order4;
  a.scale = scale; a.shift = shift; a.out = out; a.psum = psum;
  a.in_rows = (uint32_t)n_in_cap;
  a.w_bytes = (uint32_t)((uint64_t)rg.K * cin * cout * 4);           // the fragments; the pack scale's inverse sits right behind them
  a.K = rg.K; a.relu = relu ? 1 : 0; a.cap_groups = rg.cap_groups;
  a.flags = flags;
  a.res = residual;
  a.in_split = (split_io & 1) ? 1 : 0;
  if (in_absmax) {                                       // operand autoscale: max |in| -> in_absmax[1] (one memset + one launch)
    EGONN_REQUIRE(!(split_io & 1) && !gated_in2 && in_elems > 0, EGONN_ERR_INVALID, "sconv(split): operand scale on a plain fp32 input only");
    HIP_CHECK(hipMemsetAsync(in_absmax, 0, 8, stream));
    hipLaunchKernelGGL(split_absmax_kernel, dim3((unsigned)std::min<int64_t>(cdiv(in_elems, 1024), 1024)), dim3(256), 0, stream, in, in_elems, in_absmax);
    a.in_maxbits = in_absmax + 1;
  }
  a.out_split = (split_io & 2) ? 1 : 0;
  if (kparts > 1) {
    EGONN_REQUIRE(kparts <= rg.K && !gated_in2, EGONN_ERR_INVALID, "sconv(split): %d offset parts on a %d-slot map", kparts, rg.K);
    EGONN_REQUIRE(part && part_floats >= sconv_split_part_floats(rg, cout, kparts), EGONN_ERR_STATE,
                  "sconv(split): no scratch for the partial tiles of an offset-split launch");
    a.part = part;
    a.kp_n = kparts;
  }
  if (cfg == 0) cfg = sconv_split_default_cfg(cin, cout, groups_hint);
  const bool trace = cfg >= 9000;                        // 9000 + shape: the s_memtime build (tools/split_trace.py)
  int shape = trace ? cfg - 9000 : cfg;
  int parts_sel = 0;
  if (shape >= 500) { parts_sel = shape / 400; shape -= 400 * parts_sel; }
  const int ns_tot = cout / 32;
  int parts = 1;
  if (parts_sel > 0) parts = std::min(ns_tot, 1 << (parts_sel - 1));
  else if (col_parts > 0) parts = std::min(ns_tot, col_parts);      // the layer's rule (offset-split launches)
  else if (kw > 1) parts = std::max(1, ns_tot / 2);                 // in-workgroup offset parts: 64 columns per workgroup, whatever the
                                                                    // capacity (the choice of KW must not depend on the batch)
  else if (ns_tot >= 2 && cdiv(groups_hint, 4) < 700) {  // less than one round of the chip: two column parts per task
    parts = 2;                                           // (measured, profiles/r03i_colparts.txt: L4 128->128 101 / 80 / 93 us
  }                                                      //  with 1 / 2 / 4 parts, 64->128 56 / 45 / 51, L3 64->64 43 / 41; round 5, fp16 kernels:
                                                         //  4 parts change neither the layers (profiles/r05g_parts4.txt) nor scans/s)
  const int nsw = ns_tot / parts;
  if (gated_in2) {
    EGONN_REQUIRE(cin == 32 && cout == 32 && gated_gate && B >= 1 && !a.in_split && cfg == 142, EGONN_ERR_INVALID,
                  "sconv(split): the gated input exists for the 32->32 plan only");
    a.in2 = gated_in2; a.gate = gated_gate; a.B = B;
    return launch_split<32, 32, 4, 1, false, true>(a, groups_hint, stream);
  }
#define EGONN_SP_KW(CI, CO, NWW, NSWW, KWW)                                                               \
  if constexpr (SplitGeom<NSWW, NWW, false, KWW>::LDS_BYTES <= 160 * 1024) {                              \
    if (kw == KWW && a.kp_n > 1) return launch_split<CI, CO, NWW, NSWW, false, false, true, KWW>(a, groups_hint, stream);  \
    if (kw == KWW) return launch_split<CI, CO, NWW, NSWW, false, false, false, KWW>(a, groups_hint, stream); \
  }
#define EGONN_SP_LOCK1(CI, CO, NWW, NSWW)                                                                 \
  if (cin == CI && cout == CO && shape == 100 + NWW * 10 + 2 && nsw == NSWW && !trace) {                  \
    if constexpr (NWW == 4 && CI >= 64) {                                                                 \
      EGONN_SP_KW(CI, CO, NWW, NSWW, 2) EGONN_SP_KW(CI, CO, NWW, NSWW, 3) EGONN_SP_KW(CI, CO, NWW, NSWW, 4) \
      if (a.kp_n > 1) return launch_split<CI, CO, NWW, NSWW, false, false, true>(a, groups_hint, stream); \
    }                                                                                                     \
    EGONN_REQUIRE(a.kp_n == 1 && kw <= 1, EGONN_ERR_INVALID, "sconv(split): no offset-split instantiation for %d->%d (parts %d, kw %d)", cin, cout, a.kp_n, kw); \
    return launch_split<CI, CO, NWW, NSWW>(a, groups_hint, stream);                                       \
  }
#define EGONN_SP_LOCK(CI, CO, NWW)                                                                        \
  EGONN_SP_LOCK1(CI, CO, NWW, (CO / 32))                                                                  \
  if constexpr (CO >= 64) { EGONN_SP_LOCK1(CI, CO, NWW, (CO / 64)) }                                      \
  if constexpr (CO >= 128) { EGONN_SP_LOCK1(CI, CO, NWW, (CO / 128)) }
#define EGONN_SP_LOCK_TRACE(CI, CO, NWW)                                                                  \
  if (cin == CI && cout == CO && shape == 100 + NWW * 10 + 2 && trace) {                                  \
    a.trace = g_sconv_trace;                                                                              \
    return launch_split<CI, CO, NWW, (CO / 32), true>(a, groups_hint, stream);                            \
  }
  EGONN_SP_LOCK_TRACE(32, 32, 4) EGONN_SP_LOCK_TRACE(32, 32, 8) EGONN_SP_LOCK_TRACE(64, 64, 4)
  if (cin == 128 && cout == 128 && shape == 142 && trace) {              // the small maps' shape: two column parts
    a.trace = g_sconv_trace;
    return launch_split<128, 128, 4, 2, true>(a, groups_hint, stream);
  }
#define EGONN_SP_PLAN(CI, CO) EGONN_SP_LOCK(CI, CO, 8) EGONN_SP_LOCK(CI, CO, 4)
  EGONN_SP_PLAN(32, 32)
  EGONN_SP_PLAN(32, 64)
  EGONN_SP_PLAN(64, 64)
  EGONN_SP_PLAN(64, 128)
  EGONN_SP_PLAN(128, 128)
  EGONN_SP_PLAN(64, 32)
  EGONN_SP_PLAN(128, 64)
#undef EGONN_SP_PLAN
#undef EGONN_SP_LOCK
#undef EGONN_SP_LOCK1
#undef EGONN_SP_LOCK_TRACE
  set_error("sconv(split): no instantiation for %d->%d cfg %d", cin, cout, cfg);
  return EGONN_ERR_INVALID;
}

int sconv_split_default_cfg(int cin, int cout, int64_t groups_hint) {
  (void)cin; (void)cout; (void)groups_hint;
  static const int env_cfg = [] {                        // EGONN_SPLIT_CFG: measurement override
    const char* e = getenv("EGONN_SPLIT_CFG");
    return e ? atoi(e) : 0;
  }();
  return env_cfg ? env_cfg : 142;                        // lock-step kernel, workgroups of 4 waves
}

}  // namespace egonn
