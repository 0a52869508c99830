// The resident tail of the EgoNN graph: levels 5-7 of MinkTrunk + MinkHead (global) + descriptor decoder + pooling in ONE launch.
//
// Replaces, for fp32 feature maps, the 26 launches of
//   models/minkgl.py:136-153   convs[5..7] (k=2,s=2) + bn + ReLU, blocks[5..7] (ECABasicBlock, layers/eca_block.py:56-73)
//   models/minkgl.py:46-60     MinkHead: conv1x1[7] -> tconv[7] -> + conv1x1[6] -> tconv[6] -> + conv1x1[5]
//   models/minkgl.py:207-225   global_descriptor_decoder (Linear 128 -> 192, ReLU, Linear 192 -> 256)
//   layers/pooling.py:29-86    GeM / MAC / SPoC over the rows of every scan
//
// Why one kernel.  At batch 16 these levels hold 223 / 93 / 43 rows per scan; every one of their launches is a chain of dependent
// round trips (group count -> tables -> rows -> BN vectors -> stores) around < 1 us of matrix work, and each wave of the per-layer
// kernels re-reads the whole 128x128 kernel of every offset it touches (0.055 of the HBM roof, 1.6 x the algorithmic bytes,
// profiles/r04z_*).  Here the maps are small and the WEIGHTS are the big operand, so the decomposition is weight-stationary:
//
//   * a CLUSTER of 8 workgroups owns one scan; workgroup g of the cluster owns the 16 output columns [16g, 16g+16) of every
//     stage (column tiles g, g+8 of the 192- and 256-wide decoder layers) and ALL rows of the scan.  Block index = 8*scan + g,
//     so (observed placement, block b on XCD b % 8) XCD g streams only the eighth of every kernel that belongs to column tile g:
//     2.3 MB per XCD over the whole tail, L2-resident, shared by the 16 scans of the batch.  Placement is a speed matter only.
//   * a stage = out[rows][16g..] = act((sum_ops sum_k in[nbr[row][k]] @ W[k][:, 16g..]) * scale + shift).  The scan's input rows
//     are split ONCE per workgroup into fp16 hi / lo planes in LDS (the two-way split of sconv_split.hip: fp32 x fp32 = three
//     fp16 MFMA products, fp32 accumulate; weights pre-scaled by a power of two per stage), in rounds of (<= 613 rows) x
//     (32 / 64 / 128 channels) that fit 2 x 48 KB;
//     the 8 waves split the K * channel-block ITEMS of a round among themselves (item i -> wave i % 8): every W fragment enters
//     the CU exactly once (2 x 1 KB per item and wave by LDS-DMA into the wave's ring, one item group ahead), and a wave walks
//     the <= 16 row tiles of the window for its item (row fragments: 2 x ds_read_b128 per tile, 3 MFMAs).  No barrier inside a round.
//   * the 8 partial accumulators of a tile are summed by a fixed three-round exchange tree through LDS that leaves two finished
//     tiles in every wave (all waves run the epilogue): BN scale/shift or bias, ReLU, 16-byte stores, per-column sums (ECA
//     pooling, layers/eca_block.py:21-36) or the GeM / MAC / SPoC reduction of the descriptor.
//   * between stages the 8 workgroups of a scan exchange their column slices through memory: WRITE-THROUGH (sc1) stores, every
//     storing wave drains (s_waitcnt vmcnt(0)), one lane publishes the workgroup's epoch flag; consumers poll the 8 flags of
//     their cluster with relaxed agent-scope loads and read the payload with sc1 loads (cdna_hip_programming.md Guideline 16,
//     form R1).  Flags are monotonic across launches (every launch advances every flag of a cluster by the same count; a
//     workgroup reads its own flag as the base), so nothing has to be zeroed per launch or per graph replay.  Every spin is
//     bounded; a timeout sets bit 2 of the plan's flag word (egonn_plan_status reports it) instead of hanging.
//
// Summation order of an output element: fixed by (rows of the scan at the input level, channel count) only — not by the batch,
// the other scans, eager vs graph execution or the placement: results are bitwise reproducible and batch-invariant.  They
// differ from the per-layer kernels by summation order only (<= 3e-6 of the largest output, tests/test_gpu_tail.py).
#include <stdlib.h>
#include <algorithm>

#include "common.h"
#include <hip/hip_ext.h>
#include "kernels.h"

namespace egonn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

static constexpr int TL_THREADS = 512, TL_WAVES = 8;
static constexpr int TL_PLANE = 49152;                       // bytes of one fp16 plane (hi | lo; TL_PLANE < 65536: ds offsets)
static constexpr int TL_OFF_W = 2 * TL_PLANE;                // W ring: [8 waves][2 slots][2 parts][1 KB], filled by LDS-DMA
static constexpr int TL_OFF_TBL = TL_OFF_W + TL_WAVES * 2 * 2048;   // [K][16 slots][16 tiles] u16: input row of the window's (tile, offset, slot)
                                                                    // relative to the scan's first row at the input level, 0xFFFF = absent
static constexpr int TL_TBL_ROW = 48;                        // bytes of one (offset, slot) row of the staged table: 16 tiles x u16 + 16
                                                             // (12-dword stride: the 16 rows a b128 lane group reads hit disjoint banks;
                                                             // int32 entries at a 64-byte stride were a 4-way conflict, 64 of the ~170
                                                             // LDS cycles of an item)
static constexpr int TL_TBL_BYTES = 27 * 16 * TL_TBL_ROW;
static constexpr int TL_OFF_PERM = TL_OFF_TBL + TL_TBL_BYTES;   // [16 tiles][16] output rows
static constexpr int TL_OFF_MISC = TL_OFF_PERM + 1024;          // column sums [8 waves][16], pooled [16], gate [16], scratch
static constexpr int TL_LDS = TL_OFF_MISC + 1024;               // 160 768 bytes: one workgroup per CU
static constexpr int TL_SC1 = 16;                            // buffer aux bits: sc1 (write-through store / L1-bypassing load)
// rows a round can stage (+ one all-zero row) per channels-per-round: (49152 / (2 * ch + 16)) - 1
static constexpr int TL_CAP128 = 179, TL_CAP64 = 340, TL_CAP32 = 613;
static constexpr int TL_RU = 6;                              // staging units (8 channels of a row) per thread and round
static_assert((TL_CAP128 + 1) * 16 <= TL_RU * TL_THREADS && (TL_CAP64 + 1) * 8 <= TL_RU * TL_THREADS && (TL_CAP32 + 1) * 4 <= TL_RU * TL_THREADS, "staging loop covers a round");
static_assert((TL_CAP128 + 1) * (2 * 128 + 16) <= TL_PLANE && (TL_CAP64 + 1) * (2 * 64 + 16) <= TL_PLANE && (TL_CAP32 + 1) * (2 * 32 + 16) <= TL_PLANE, "plane");
static_assert(TL_WAVES * 8 * 64 * 16 <= 2 * TL_PLANE, "the reduction tree reuses the planes");
static_assert(TL_LDS <= 160 * 1024, "LDS budget");

// ------------------------------------------------------------------ weight packing
// fp32 x fp32 on the fp16 matrix pipe as in sconv_split.hip: x = hi + lo (fp16, round to nearest) leaves 2^-22 |x|; a*w = ah*wh +
// ah*wl + al*wh (three v_mfma_f32_16x16x32_f16, fp32 accumulate).  Weights are scaled by a power of two s per SCALE GROUP (the
// kernels one stage accumulates into the same registers: max |W| -> [2^13, 2^14)) so that their low parts stay normal fp16 numbers;
// 1/s is applied exactly in the stage's epilogue.
// W [K][cin][cout] (ME kernel layout; out_in: nn.Linear [cout][cin], K = 1) -> [ct][k][cb][part][lane][e] fp16,
//   = part( s * W[k][32 cb + 8 (lane >> 4) + e][16 ct + (lane & 15)] ),  part 0 hi, 1 lo
// The slice of column tile ct is one contiguous block of K * cin/32 * 2 KB; every 1 KB piece is one lane-linear MFMA operand.
// trailer (4 words per scale group): [0] float 1/s, [1] bits of max |W| over the group.
__global__ void tail_absmax_kernel(const float* __restrict__ W, int64_t n, uint32_t* __restrict__ trailer) {
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
static __device__ inline float tl_scale_of(uint32_t maxbits) {          // power of two s with s * max in [2^13, 2^14)
  const int e = (int)(maxbits >> 23) - 127;
  const int se = min(max(13 - e, -100), 100);
  return __uint_as_float((uint32_t)(se + 127) << 23);
}
__global__ void pack_tail_weights_kernel(const float* __restrict__ W, int K, int cin, int cout, int out_in,
                                         uint16_t* __restrict__ out, uint32_t* __restrict__ trailer) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float sc = tl_scale_of(trailer[1]);
  if (t == 0) reinterpret_cast<float*>(trailer)[0] = 1.f / sc;
  if (t >= (int64_t)K * cin * cout * 2) return;
  const int ncb = cin / 32;
  int64_t r = t;
  const int e = (int)(r & 7); r >>= 3;
  const int lane = (int)(r & 63); r >>= 6;
  const int part = (int)(r & 1); r >>= 1;
  const int cb = (int)(r % ncb); r /= ncb;
  const int k = (int)(r % K); r /= K;
  const int ct = (int)r;
  const int ci = 32 * cb + 8 * (lane >> 4) + e, co = 16 * ct + (lane & 15);
  const float v = sc * (out_in ? W[(int64_t)co * cin + ci] : W[((int64_t)k * cin + ci) * cout + co]);
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)(v - (float)hi);
  out[t] = __builtin_bit_cast(uint16_t, part == 0 ? hi : lo);
}

int tail_weights_absmax(const float* W, int64_t n, void* trailer, hipStream_t stream) {
  hipLaunchKernelGGL(tail_absmax_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n, 256), 256)), dim3(256), 0, stream, W, n,
                     reinterpret_cast<uint32_t*>(trailer));
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}
int pack_tail_weights(const float* W, int K, int cin, int cout, int out_in, void* out, void* trailer, hipStream_t stream) {
  EGONN_REQUIRE(cin % 32 == 0 && cout % 16 == 0 && (!out_in || K == 1), EGONN_ERR_INVALID, "tail: channel plan %d->%d", cin, cout);
  const int64_t n = (int64_t)K * cin * cout * 2;
  hipLaunchKernelGGL(pack_tail_weights_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, W, K, cin, cout, out_in,
                     reinterpret_cast<uint16_t*>(out), reinterpret_cast<uint32_t*>(trailer));
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------ device helpers
namespace {

struct TlOp {
  const float* in;         // input map [rows][cin]
  const int32_t* tbl;      // [groups][K][16] input row of every (group, offset, slot), -1 = absent
  const uint16_t* W;       // pack_tail_weights
  const float* inv;        // 1 / (weight scale of the op's scale group)
  const int32_t* in_boff;  // per-scan row offsets of the input level
  int K, cin, in_cap, lin; // in_cap: rows the input map holds; lin: input level (row count = cnt[lin])
  int maskbits;            // 1: offset k is present in a tile iff bit k of its group mask; 0: iff the tile has real rows
};
struct TlDesc {
  int kind;                // 0 = convolution stage, 1 = ECA gate + residual + ReLU on the workgroup's own columns
  int nops;
  TlOp op[2];
  TailMap til;             // output tiling (row groups of the output level)
  const float *scale, *shift;
  int relu, ncoltiles, ostride, out_cap, pool, lout;   // pool: 0 none, 1 per-column sums -> a.sums, 2 descriptor pooling -> a.out_global
  float* out;
  // kind 1
  const float *t2, *res, *eca_w;
  int eca_k;
};

__device__ inline float tl_row16_sum(float v) {              // sum over the 16 lanes of a DPP row (the 16 rows of a tile)
  int x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));
  x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));
  x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));
  x = __builtin_bit_cast(int, v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true));
  return v;
}
__device__ inline float tl_row16_max(float v) {
  int x = __builtin_bit_cast(int, v);
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false)));
  x = __builtin_bit_cast(int, v);
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false)));
  x = __builtin_bit_cast(int, v);
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false)));
  x = __builtin_bit_cast(int, v);
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false)));
  return v;
}

// fp32 x 8 -> (hi, lo) fp16 x 8, round to nearest even (as sconv_split.hip split8h)
__device__ inline void tl_split8(const f32x4& a0, const f32x4& a1, f16x8_t& hi, f16x8_t& lo) {
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

__device__ inline uint32_t tl_ld_flag(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// stage decode: s = 0..11: level 5 + s/4, phase s%4 (k2s2 | conv1 | conv2 | gate + residual); 12..16: head
__device__ inline TlDesc tl_decode(const TailArgs& a, int s) {
  TlDesc d;
  d.kind = 0; d.nops = 1; d.scale = nullptr; d.shift = nullptr; d.relu = 0; d.ncoltiles = 8; d.ostride = 128; d.pool = 0;
  d.out = nullptr; d.t2 = nullptr; d.res = nullptr; d.eca_w = nullptr; d.eca_k = 0; d.out_cap = 0; d.lout = 5;
  d.op[1].in = nullptr; d.op[1].tbl = nullptr; d.op[1].W = nullptr; d.op[1].in_boff = nullptr; d.op[1].inv = nullptr;
  d.op[1].K = 0; d.op[1].cin = 0; d.op[1].in_cap = 0; d.op[1].lin = 0; d.op[1].maskbits = 0;
  TlOp& o = d.op[0];
  o.cin = 128; o.maskbits = 1;
  o.inv = a.w_inv + 4 * s;          // one scale group per stage (both ops of a head stage share it)
  if (s < 12) {
    const int li = s >> 2, ph = s & 3, lv = 5 + li;
    d.lout = lv; d.out_cap = a.cap[lv];
    if (ph == 0) {
      o.in = li == 0 ? a.x4 : a.x[li - 1]; o.lin = lv - 1; o.tbl = a.rg8[li].snbr; o.K = 8; o.W = a.w_k2[li];
      d.til = a.rg8[li]; d.scale = a.bn_s[li]; d.shift = a.bn_h[li]; d.relu = 1; d.out = a.y[li];
    } else if (ph == 1) {
      o.in = a.y[li]; o.lin = lv; o.tbl = a.rg27[li].snbr; o.K = 27; o.W = a.w_c1[li];
      d.til = a.rg27[li]; d.scale = a.n1_s[li]; d.shift = a.n1_h[li]; d.relu = 1; d.out = a.t1[li];
    } else if (ph == 2) {
      o.in = a.t1[li]; o.lin = lv; o.tbl = a.rg27[li].snbr; o.K = 27; o.W = a.w_c2[li];
      d.til = a.rg27[li]; d.scale = a.n2_s[li]; d.shift = a.n2_h[li]; d.relu = 0; d.out = a.t2[li]; d.pool = 1;
    } else {
      d.kind = 1; d.til = a.rg27[li]; d.t2 = a.t2[li]; d.res = a.y[li]; d.out = a.x[li]; d.eca_w = a.eca_w[li]; d.eca_k = a.eca_k[li];
      o.in = nullptr; o.tbl = nullptr; o.W = nullptr; o.K = 0; o.lin = lv;
    }
  } else if (s == 12) {          // g7 = x7 @ W1x1[7]
    o.in = a.x[2]; o.lin = 7; o.tbl = a.rg27[2].perm; o.K = 1; o.W = a.w_1x1[2]; o.maskbits = 0;
    d.til = a.rg27[2]; d.out = a.g7; d.lout = 7; d.out_cap = a.cap[7];
  } else if (s == 13 || s == 14) {   // g6 = x6 @ W1x1[6] + tconv7(g7) ; g5 = x5 @ W1x1[5] + tconv6(g6)
    const int j = s - 13, lv = 6 - j;          // j = 0: onto level 6, 1: onto level 5
    o.in = a.x[lv - 5]; o.lin = lv; o.tbl = a.rgT[j].perm; o.K = 1; o.W = a.w_1x1[lv - 5]; o.maskbits = 0;
    TlOp& t = d.op[1];
    t.in = j == 0 ? a.g7 : a.g6; t.lin = lv + 1; t.tbl = a.rgT[j].snbr; t.K = 8; t.W = a.w_t[j]; t.cin = 128; t.maskbits = 1;
    d.nops = 2; d.til = a.rgT[j]; d.out = j == 0 ? a.g6 : a.g5; d.lout = lv; d.out_cap = a.cap[lv];
  } else if (s == 15) {          // gh = relu(g5 @ W0^T + b0)
    o.in = a.g5; o.lin = 5; o.tbl = a.rg27[0].perm; o.K = 1; o.W = a.w_m0; o.maskbits = 0;
    d.til = a.rg27[0]; d.shift = a.b0; d.relu = 1; d.ncoltiles = 12; d.ostride = 192; d.out = a.gh; d.lout = 5; d.out_cap = a.cap[5];
  } else {                       // descriptor rows = gh @ W1^T + b1, pooled per scan
    o.in = a.gh; o.lin = 5; o.cin = 192; o.tbl = a.rg27[0].perm; o.K = 1; o.W = a.w_m1; o.maskbits = 0;
    d.til = a.rg27[0]; d.shift = a.b1; d.ncoltiles = 16; d.ostride = 256; d.out = nullptr; d.pool = 2; d.lout = 5; d.out_cap = a.cap[5];
  }
  d.op[0].in_boff = a.boff[d.op[0].lin];
  d.op[0].in_cap = a.cap[d.op[0].lin];
  d.op[1].in_boff = a.boff[d.op[1].lin];
  d.op[1].in_cap = a.cap[d.op[1].lin];
  return d;
}

}  // namespace

// ------------------------------------------------------------------ the kernel
template <bool TRACE>    // TRACE: measurement build (tools/tail_trace.py), s_memtime stamps per stage; the release build has none
__global__ __launch_bounds__(TL_THREADS) void tail_kernel(const TailArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g4 = lane >> 4;
  const int b = blockIdx.x >> 3, g = blockIdx.x & 7;
  char* const s_tbl = smem + TL_OFF_TBL;
  int32_t* const s_perm = reinterpret_cast<int32_t*>(smem + TL_OFF_PERM);
  float* const s_csum = reinterpret_cast<float*>(smem + TL_OFF_MISC);          // [8][16]
  float* const s_pool = s_csum + 128;                                          // [16] accumulated over the windows of a scan
  float* const s_gate = s_csum + 160;                                          // [16]
  int32_t* const s_flag = reinterpret_cast<int32_t*>(s_csum + 192);           // [0] sync result

  uint32_t* const flags = a.flags + (size_t)b * 8;
  const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)tl_ld_flag(flags + g));
  uint32_t epoch = 0;               // stages this cluster has published
  bool dead = false;                // a wait timed out: keep going without waiting (results are flagged invalid)

  // wait until every workgroup of the cluster has published `epoch` stages
  auto cluster_wait = [&]() {
    if (epoch == 0) return;
    if (wave == 0) {
      int ok = 1;
      if (!dead) {
        const uint32_t want = base + epoch;
        ok = 0;
        for (int spin = 0; spin < (1 << 21); ++spin) {
          const uint32_t v = lane < 8 ? tl_ld_flag(flags + lane) : want;
          if (__all((int32_t)(v - want) >= 0)) { ok = 1; break; }
          __builtin_amdgcn_s_sleep(2);
        }
      }
      if (lane == 0) s_flag[0] = ok;
    }
    __syncthreads();
    if (!__builtin_amdgcn_readfirstlane(s_flag[0]) && !dead) {
      dead = true;
      if (tid == 0) atomicOr(a.err, 4);
    }
    __syncthreads();
  };
  // every wave has drained its write-through stores; one lane publishes the stage
  auto cluster_publish = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++epoch;
    if (tid == 0) __hip_atomic_store(flags + g, base + epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  // measurement hook (tools/tail_trace.py): per (workgroup, stage) s_memtime stamps of wave 0; null = off
  // (the stamps live 2^19 entries into the buffer: the traced builds of other kernels of the same forward write at its start)
  unsigned long long* const trp = (TRACE && a.trace) ? a.trace + ((size_t)1 << 19) + (size_t)blockIdx.x * 17 * 8 : nullptr;
  auto now = [] { return (unsigned long long)__builtin_amdgcn_s_memtime(); };
  unsigned long long tr[TRACE ? 8 : 1];

  const int nstage = a.do_head ? 17 : 12;
  const void* tbl_cached = nullptr;   // table currently staged in s_tbl (conv1 / conv2 of a block share it)
  int tbl_cached_base = -1, tbl_cached_row0 = -1;

  for (int s = 0; s < nstage; ++s) {
    const TlDesc d = tl_decode(a, s);
    const int n_out_lvl = __builtin_amdgcn_readfirstlane(min(a.cnt[d.lout], a.cap[d.lout]));
    if constexpr (TRACE) {
#pragma unroll
      for (int q = 0; q < 8; ++q) tr[q] = 0;
      tr[0] = now();
    }
    auto trace_out = [&]() {
      if constexpr (TRACE) {
        if (trp && tid == 0) {
          tr[5] = now();
#pragma unroll
          for (int q = 0; q < 8; ++q) trp[s * 8 + q] = tr[q];
        }
      }
    };
    if (d.kind == 1) {
      // ---------------- ECA gate + residual + ReLU (layers/eca_block.py:21-36,66-71) on the own 16 columns
      cluster_wait();
      const int rs = __builtin_amdgcn_readfirstlane(min(a.boff[d.lout][b], n_out_lvl));
      const int re = __builtin_amdgcn_readfirstlane(min(a.boff[d.lout][b + 1], n_out_lvl));
      const int nrows = re - rs;
      if (tid < 16) {
        const int c = 16 * g + tid, pad = (d.eca_k - 1) / 2;
        float yv = 0.f;
        for (int j = 0; j < d.eca_k; ++j) {
          const int q = c + j - pad;
          if (q >= 0 && q < 128) {
            const float sum = __hip_atomic_load(a.sums + (size_t)b * 128 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float mean = nrows > 0 ? sum / (float)nrows : 0.f;
            yv += d.eca_w[j] * mean;
          }
        }
        s_gate[tid] = 1.f / (1.f + expf(-yv));
      }
      __syncthreads();
      const __amdgpu_buffer_rsrc_t r_t2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.t2), 0, a.cap[d.lout] * 512, 0x00020000);
      const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.res), 0, a.cap[d.lout] * 512, 0x00020000);
      const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(d.out, 0, a.cap[d.lout] * 512, 0x00020000);
      for (int i = tid; i < nrows * 4; i += TL_THREADS) {
        const int row = rs + (i >> 2), q = i & 3;
        const int off = row * 512 + (16 * g + 4 * q) * 4;
        const f32x4 xv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_t2, off, 0, TL_SC1));
        const f32x4 rv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_res, off, 0, TL_SC1));
        f32x4 o;
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = fmaxf(xv[u] * s_gate[4 * q + u] + rv[u], 0.f);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), r_out, off, 0, TL_SC1);
      }
      cluster_publish();
      trace_out();
      continue;
    }

    // ---------------- convolution stage
    bool synced = (s == 0);          // the first stage reads the previous launch's output
    const int G0 = __builtin_amdgcn_readfirstlane(d.til.meta[1 + b]), G1 = __builtin_amdgcn_readfirstlane(d.til.meta[2 + b]);
    const int npass = (G1 - G0 + 15) >> 4;
    const int out_rs = __builtin_amdgcn_readfirstlane(min(a.boff[d.lout][b], n_out_lvl));
    const int out_re = __builtin_amdgcn_readfirstlane(min(a.boff[d.lout][b + 1], n_out_lvl));
    for (int ct = g; ct < d.ncoltiles; ct += 8) {
      if (tid < 16) s_pool[tid] = d.pool == 2 && a.pool_mode == 2 ? -INFINITY : 0.f;
      // epilogue vectors of this column tile: in flight from here
      const int col0 = 16 * ct + 4 * g4;
      // sc carries 1/s of the stage's weight scale (a power of two: exact)
      const float winv = d.op[0].inv[0];
      f32x4 sc = (f32x4){winv, winv, winv, winv}, sh = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (d.scale) sc = *reinterpret_cast<const f32x4*>(d.scale + col0) * winv;
      if (d.shift) sh = *reinterpret_cast<const f32x4*>(d.shift + col0);
      for (int p = 0; p < npass; ++p) {
        const int gbase = G0 + 16 * p;
        const uint32_t gm_l = (gbase + l15 < G1) ? d.til.gmask[gbase + l15] : 0u;       // mask of tile (lane & 15)
        const uint32_t act_all = (uint32_t)__ballot((int)(gm_l >> 31)) & 0xFFFFu;          // tiles with real rows
        int32_t perm_v = -1;
        if (tid < 256 && gbase + (tid >> 4) < G1) perm_v = d.til.perm[(size_t)(gbase + (tid >> 4)) * 16 + (tid & 15)];

        // ---- per-op row ranges and round geometry.  An op whose table is the tiling's own row list (1x1 / Linear layers)
        // needs only the window's rows; every other op may reference any row of the scan at its input level.
        int o_rs[2], o_re[2], o_chc[2], o_cap[2];
#pragma unroll
        for (int oi = 0; oi < 2; ++oi) {
          const TlOp& op = d.op[oi];
          int rs = 0, re = 0;
          if (oi < d.nops) {
            const int n_in_lvl = __builtin_amdgcn_readfirstlane(min(a.cnt[op.lin], op.in_cap));
            rs = __builtin_amdgcn_readfirstlane(min(op.in_boff[b], n_in_lvl));
            re = __builtin_amdgcn_readfirstlane(min(op.in_boff[b + 1], n_in_lvl));
            if (!op.maskbits) { rs = min(re, rs + 256 * p); re = min(re, rs + 256); }
          }
          const int nrows = re - rs;
          o_rs[oi] = rs; o_re[oi] = re;
          if (nrows <= TL_CAP128) { o_chc[oi] = min(128, op.cin); o_cap[oi] = TL_CAP128; }
          else if (nrows <= TL_CAP64) { o_chc[oi] = 64; o_cap[oi] = TL_CAP64; }
          else { o_chc[oi] = 32; o_cap[oi] = TL_CAP32; }
        }
        // a round = (op, piece of its input rows, channel chunk); oi = 2: past the end
        struct Rnd { int oi, r0, c0; };
        auto rnd_norm = [&](Rnd r) {          // skip ops without rows
          while (r.oi < d.nops && (r.oi == 0 ? o_rs[0] >= o_re[0] : o_rs[1] >= o_re[1])) { ++r.oi; r.r0 = r.oi == 1 ? o_rs[1] : 0; r.c0 = 0; }
          if (r.oi >= d.nops) r.oi = 2;
          return r;
        };
        auto rnd_next = [&](Rnd r) {
          const int chc = r.oi == 0 ? o_chc[0] : o_chc[1], cap = r.oi == 0 ? o_cap[0] : o_cap[1];
          const int cin = r.oi == 0 ? d.op[0].cin : d.op[1].cin, re = r.oi == 0 ? o_re[0] : o_re[1];
          r.c0 += chc;
          if (r.c0 >= cin) { r.c0 = 0; r.r0 += cap; }
          if (r.r0 >= re) { ++r.oi; r.r0 = r.oi == 1 ? o_rs[1] : 0; r.c0 = 0; }
          return rnd_norm(r);
        };
        // geometry of a round
        struct Geo { const float* in; const uint16_t* W; const int32_t* tbl; int K, cin, ncb, maskbits, in_cap, nr, ncbc, nc8, sh8, stride, nitems, row0; };
        auto geo_of = [&](const Rnd& r) {
          Geo q;
          const TlOp& op = r.oi == 0 ? d.op[0] : d.op[1];
          q.in = op.in; q.W = op.W; q.tbl = op.tbl; q.K = op.K; q.cin = op.cin; q.ncb = op.cin >> 5; q.maskbits = op.maskbits;
          q.in_cap = op.in_cap;
          q.row0 = r.oi == 0 ? o_rs[0] : o_rs[1];            // table entries are stored relative to this row
          const int chc = r.oi == 0 ? o_chc[0] : o_chc[1], cap = r.oi == 0 ? o_cap[0] : o_cap[1], re = r.oi == 0 ? o_re[0] : o_re[1];
          q.nr = min(cap, re - r.r0);
          q.ncbc = min(chc, op.cin - r.c0) >> 5;
          q.nc8 = q.ncbc * 4;
          q.sh8 = 31 - __builtin_clz(q.nc8);
          q.stride = q.ncbc * 64 + 16;
          q.nitems = op.K * q.ncbc;
          return q;
        };

        f32x4 acc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // W fragments.  The items of a round are walked in GROUPS of <= 32 (one channel block of a k=3 map; up to four of a
        // k=2 / 1x1 map): slot u of this wave = item wave + 8 u of the group, u = 0..3.  Every load below is UNCONDITIONAL
        // and in a fixed order (a slot without an item loads out of range: zeros, no traffic), so that the compiler's own
        // counted s_waitcnt covers the ring — loads inside wave-uniform branches made it drain vmcnt(0) in front of every item
        // (1.8 k cycles per item, tools/tail_trace.py).
        // W fragments.  The K * ncbc items of a round are walked in GROUPS of 16 consecutive items (item = cbl * K + k); slot u
        // (0, 1) of this wave holds item 16 gi + wave + 8 u of group gi.  A fragment set (hi | lo, 2 KB) goes global ->
        // LDS by LDS-DMA into the wave's own ring slot one group ahead — also across the barriers between rounds — and is read
        // back with three ds_read_b128 when its item starts.  Neither the compiler's wait-count pass (it drained vmcnt(0) in
        // front of every tile: 1.8 k cycles per item, tools/tail_trace.py) nor a register ring filled by asm (hipcc copies such
        // registers while their loads are in flight) survives these loops, so: the DMA is an asm statement with no register
        // result, every request is UNCONDITIONAL and in a fixed order (a slot without an item requests out of range: zeros, no
        // traffic), and the waits are counted by hand — requests return in order, and the number of requests younger than a
        // slot's at its use is 2 (the other slot's refill), + 2 TL_RU = 12 row loads of the next round in the first group of a round.
        typedef __attribute__((address_space(3))) char lds_char;
        const uint32_t wslot0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_char*)(smem + TL_OFF_W + wave * 4096));
        const uint32_t smem_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(lds_char*)smem);
        auto w_dma = [&](const uint16_t* Wp, int K, int ncb, int nitems, int c0, bool valid_r, int gi, int u) {
          const int item = gi * 16 + wave + TL_WAVES * u;
          const bool valid = valid_r && item < nitems;
          const int cbl = (int)(((float)item + 0.5f) * (1.f / (float)K)), k = item - cbl * K;
          const uint64_t wb = (uint64_t)(uintptr_t)(Wp + (size_t)ct * K * ncb * 1024);
          u32x4 rs;
          rs[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wb);
          rs[1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(wb >> 32)) & 0xFFFFu;
          rs[2] = (uint32_t)__builtin_amdgcn_readfirstlane(K * ncb * 2048);
          rs[3] = 0x00020000u;
          const int so = __builtin_amdgcn_readfirstlane(valid ? (k * ncb + (c0 >> 5) + cbl) * 2048 : 0);
          const int vo = valid ? lane * 16 : (int)0x80000000u;
          const uint32_t dst = wslot0 + (uint32_t)u * 2048u;
          uint32_t keep;
          asm volatile(
              "s_mov_b32 %0, m0\n\t"
              "s_mov_b32 m0, %3\n\t"
              "s_nop 4\n\t"
              "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
              "buffer_load_dwordx4 %1, %2, %4 offen offset:1024 lds\n\t"
              "s_mov_b32 m0, %0"
              : "=&s"(keep)
              : "v"(vo), "s"(rs), "s"(dst), "s"(so)
              : "memory");
        };
        // rows of a round: 2 TL_RU x 16 bytes per thread in flight, then split into the fp16 planes
        f32x4 rv0[TL_RU], rv1[TL_RU];
        auto rows_issue = [&](const Rnd& r, const Geo& q, bool valid_r) {
          const __amdgpu_buffer_rsrc_t r_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.in), 0, q.in_cap * q.cin * 4, 0x00020000);
#pragma unroll
          for (int u = 0; u < TL_RU; ++u) {
            const int unit = tid + TL_THREADS * u;
            const int row = unit >> q.sh8, c8 = unit & (q.nc8 - 1);
            const int voff = (valid_r && row < q.nr) ? ((r.r0 + row) * q.cin + r.c0 + 8 * c8) * 4 : (int)0x80000000u;
            rv0[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_in, voff, 0, TL_SC1));
            rv1[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_in, voff, 16, TL_SC1));
          }
        };
        auto rows_commit = [&](const Geo& q) {
#pragma unroll
          for (int u = 0; u < TL_RU; ++u) {
            const int unit = tid + TL_THREADS * u;
            const int row = unit >> q.sh8, c8 = unit & (q.nc8 - 1);
            if (row <= q.nr) {                          // row nr: the all-zero row absent neighbours read
              f16x8_t hi, lo;
              tl_split8(rv0[u], rv1[u], hi, lo);
              char* dst = smem + row * q.stride + c8 * 16;
              *reinterpret_cast<f16x8_t*>(dst) = hi;
              *reinterpret_cast<f16x8_t*>(dst + TL_PLANE) = lo;
            }
          }
        };
        // the window's table of an op -> registers (16-byte pieces: 4 slots of one (tile, offset)) -> LDS as [k][slot][tile]
        auto tbl_stage = [&](const Geo& q) {
          const int ppt = q.K * 4;                      // pieces per tile
          const float inv = 1.f / (float)ppt;
          int4 tv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int j = tid + TL_THREADS * u;
            const int t = (int)(((float)j + 0.5f) * inv);
            tv[u] = make_int4(-1, -1, -1, -1);
            if (t < 16 && gbase + t < G1) tv[u] = reinterpret_cast<const int4*>(q.tbl + (size_t)(gbase + t) * q.K * 16)[j - t * ppt];
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int j = tid + TL_THREADS * u;
            const int t = (int)(((float)j + 0.5f) * inv);
            if (t < 16) {
              const int rem = j - t * ppt, k = rem >> 2, s0 = (rem & 3) * 4;
              uint16_t* dst = reinterpret_cast<uint16_t*>(s_tbl + (k * 16 + s0) * TL_TBL_ROW) + t;
              // (a scan with more than 65 534 rows at the input level does not fit the 16-bit entries: flagged like a timed-out
              //  wait — bit 2 of the plan's flag word, egonn_plan_status — instead of silently reading a wrong row)
              auto rel = [&](int32_t row) {
                if (row - q.row0 > 0xFFFE) atomicOr(a.err, 4);
                return (uint16_t)(row < 0 ? 0xFFFF : min(row - q.row0, 0xFFFE));
              };
              dst[0] = rel(tv[u].x); dst[TL_TBL_ROW / 2] = rel(tv[u].y); dst[TL_TBL_ROW] = rel(tv[u].z); dst[3 * TL_TBL_ROW / 2] = rel(tv[u].w);
            }
          }
        };

        // ---- prologue: everything that does not depend on the cluster is requested before the wait
        Rnd cur = rnd_norm(Rnd{0, o_rs[0], 0});
        Geo cg = geo_of(cur.oi < 2 ? cur : Rnd{0, 0, 0});
#pragma unroll
        for (int u = 0; u < 2; ++u) w_dma(cg.W, cg.K, cg.ncb, cg.nitems, cur.c0, cur.oi < 2, 0, u);
        __syncthreads();                 // previous window / stage: readers of s_perm, s_tbl and the planes are done
        if (tid < 256) s_perm[tid] = perm_v;
        if (cur.oi < 2 && (cg.tbl != tbl_cached || gbase != tbl_cached_base || cg.row0 != tbl_cached_row0)) {
          tbl_stage(cg);
          tbl_cached = cg.tbl;
          tbl_cached_base = gbase;
          tbl_cached_row0 = cg.row0;
        }
        if (!synced) { cluster_wait(); synced = true; if constexpr (TRACE) tr[1] = now(); }
        rows_issue(cur, cg, cur.oi < 2);

        while (cur.oi < 2) {
          unsigned long long t_a = 0, t_b = 0;
          if constexpr (TRACE) t_a = now();
          rows_commit(cg);
          __syncthreads();
          if constexpr (TRACE) { t_b = now(); tr[2] += t_b - t_a; tr[6] += 1; }
          const Rnd nxt = rnd_next(cur);
          const bool nv = nxt.oi < 2;
          const Geo ng = geo_of(nv ? nxt : cur);
          rows_issue(nxt, ng, nv);                       // in flight while this round computes
          const bool new_tbl = nv && ng.tbl != cg.tbl;

          // ---- the groups of this round
          const int ngrp = (cg.nitems + 15) >> 4;
          const float invK = 1.f / (float)cg.K;
          for (int gi = 0; gi < ngrp; ++gi) {
            const bool last = gi + 1 == ngrp;             // the refills fetch the first group of the next round
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int item = gi * 16 + wave + TL_WAVES * u;
              unsigned long long t_w = 0;
              if constexpr (TRACE) t_w = now();
              if (gi == 0) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
              else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
              if constexpr (TRACE) tr[7] += now() - t_w;
              if (item < cg.nitems) {
                const int cbl = (int)(((float)item + 0.5f) * invK), k = item - cbl * cg.K;
                const char* wp = smem + TL_OFF_W + (wave * 2 + u) * 2048 + lane * 16;
                const f16x8_t wh = *reinterpret_cast<const f16x8_t*>(wp);
                const f16x8_t wl = *reinterpret_cast<const f16x8_t*>(wp + 1024);
                // the 16 input rows of this lane's slot (one per tile) -> LDS offsets of their plane rows; an absent
                // neighbour (-1), a row outside this round's piece and every slot of a tile without the offset read the zero row
                const uint4* tp = reinterpret_cast<const uint4*>(s_tbl + (k * 16 + l15) * TL_TBL_ROW);
                const uint4 e0 = tp[0], e1 = tp[1];
                const uint32_t ew[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
                const int rel0 = cur.r0 - cg.row0;             // first row of this round's piece, relative like the entries
                const uint32_t am = cg.maskbits ? ((uint32_t)__ballot((gm_l >> k) & 1u) & 0xFFFFu) : act_all;
                const uint32_t abase = smem_lds + (uint32_t)(cbl * 64 + g4 * 16);
                // four tiles at a time.  The eight fragment reads of a quad (hi | lo plane of four rows) are ONE asm statement
                // that ends with its wait: left to the compiler, the lo-plane reads reused one register quadruple and were
                // serialised behind the products (five LDS round trips per quad: ~1 500 cycles per item whatever the tile
                // count, tools/tail_trace.py).  While this wave waits for its reads the other wave of the SIMD feeds the
                // matrix pipe; the three products of the four tiles are interleaved (four independent accumulator chains).
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                  if ((am >> (4 * qd)) & 0xFu) {
                    uint32_t ad[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                      const int t = 4 * qd + j;
                      const int loc = (int)((t & 1) ? (ew[t >> 1] >> 16) : (ew[t >> 1] & 0xFFFFu)) - rel0;
                      const int idx = (uint32_t)loc < (uint32_t)cg.nr ? loc : cg.nr;
                      ad[j] = abase + (uint32_t)(idx * cg.stride);
                    }
                    f32x4 rh0, rh1, rh2, rh3, rl0, rl1, rl2, rl3;
                    asm volatile(
                        "ds_read_b128 %0, %8\n\t"
                        "ds_read_b128 %1, %9\n\t"
                        "ds_read_b128 %2, %10\n\t"
                        "ds_read_b128 %3, %11\n\t"
                        "ds_read_b128 %4, %8 offset:49152\n\t"
                        "ds_read_b128 %5, %9 offset:49152\n\t"
                        "ds_read_b128 %6, %10 offset:49152\n\t"
                        "ds_read_b128 %7, %11 offset:49152\n\t"
                        "s_waitcnt lgkmcnt(0)"
                        : "=&v"(rh0), "=&v"(rh1), "=&v"(rh2), "=&v"(rh3), "=&v"(rl0), "=&v"(rl1), "=&v"(rl2), "=&v"(rl3)
                        : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3])
                        : "memory");
                    static_assert(TL_PLANE == 49152, "the lo-plane offset of the asm above");
                    const f16x8_t ah[4] = {__builtin_bit_cast(f16x8_t, rh0), __builtin_bit_cast(f16x8_t, rh1),
                                           __builtin_bit_cast(f16x8_t, rh2), __builtin_bit_cast(f16x8_t, rh3)};
                    const f16x8_t al[4] = {__builtin_bit_cast(f16x8_t, rl0), __builtin_bit_cast(f16x8_t, rl1),
                                           __builtin_bit_cast(f16x8_t, rl2), __builtin_bit_cast(f16x8_t, rl3)};
                    f32x4 c[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah[j], acc[4 * qd + j], 0, 0, 0);   // small terms first
#pragma unroll
                    for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al[j], c[j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[4 * qd + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah[j], c[j], 0, 0, 0);
                  }
                }
              }
              if (last) w_dma(ng.W, ng.K, ng.ncb, ng.nitems, nxt.c0, nv, 0, u);
              else w_dma(cg.W, cg.K, cg.ncb, cg.nitems, cur.c0, true, gi + 1, u);
            }
          }
          if constexpr (TRACE) tr[3] += now() - t_b;
          __syncthreads();               // every wave is done with the planes (and, for a new op, with the table)
          if (new_tbl) {
            tbl_stage(ng);
            tbl_cached = ng.tbl;
            tbl_cached_base = gbase;
            tbl_cached_row0 = ng.row0;
          }
          cur = nxt;
          cg = ng;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ring's last (out-of-range) requests
        if (!synced) { cluster_wait(); synced = true; }    // (a scan without input rows still takes part in the protocol)

        // ---- sum the 8 partial accumulators of every tile: three exchange rounds, two finished tiles per wave
        unsigned long long t_r = 0;
        if constexpr (TRACE) t_r = now();
        f32x4 k8[8], k4[4], fin[2];
        f32x4* const red = reinterpret_cast<f32x4*>(smem);
        {
          const bool h = (wave >> 2) & 1;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            k8[i] = h ? acc[8 + i] : acc[i];
            red[(wave * 8 + i) * 64 + lane] = h ? acc[i] : acc[8 + i];
          }
          __syncthreads();
#pragma unroll
          for (int i = 0; i < 8; ++i) k8[i] += red[((wave ^ 4) * 8 + i) * 64 + lane];
          const bool q = (wave >> 1) & 1;
          __syncthreads();
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            k4[i] = q ? k8[4 + i] : k8[i];
            red[(wave * 4 + i) * 64 + lane] = q ? k8[i] : k8[4 + i];
          }
          __syncthreads();
#pragma unroll
          for (int i = 0; i < 4; ++i) k4[i] += red[((wave ^ 2) * 4 + i) * 64 + lane];
          const bool r = wave & 1;
          __syncthreads();
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            fin[i] = r ? k4[2 + i] : k4[i];
            red[(wave * 2 + i) * 64 + lane] = r ? k4[i] : k4[2 + i];
          }
          __syncthreads();
#pragma unroll
          for (int i = 0; i < 2; ++i) fin[i] += red[((wave ^ 1) * 2 + i) * 64 + lane];
        }
        const int tbase = 8 * ((wave >> 2) & 1) + 4 * ((wave >> 1) & 1) + 2 * (wave & 1);

        // ---- epilogue of this wave's two tiles
        const __amdgpu_buffer_rsrc_t r_out =
            __builtin_amdgcn_make_buffer_rsrc(d.out, 0, d.out ? d.out_cap * d.ostride * 4 : 0, 0x00020000);
        const float gp = (d.pool == 2 && a.pool_mode == 1) ? __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.gem_p[0]))) : 1.f;
        const bool pmax = d.pool == 2 && a.pool_mode == 2;
        float cs[4] = {pmax ? -INFINITY : 0.f, pmax ? -INFINITY : 0.f, pmax ? -INFINITY : 0.f, pmax ? -INFINITY : 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int32_t row = s_perm[(tbase + j) * 16 + l15];
          f32x4 v = fin[j];
          v = v * sc + sh;
          if (d.relu) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = fmaxf(v[u], 0.f);
          }
          if (row >= 0 && d.out)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r_out, (row * d.ostride + col0) * 4, 0, TL_SC1);
          if (d.pool) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              float x = v[u];
              // GeM (layers/pooling.py:82-86): clamp(x, eps)^p on the hardware log2 / exp2 (~1e-7 relative for the terms that
              // carry the sum); the final ^(1/p) of the mean uses powf
              if (d.pool == 2 && a.pool_mode == 1) x = __builtin_amdgcn_exp2f(gp * __builtin_amdgcn_logf(fmaxf(x, 1e-6f)));
              if (pmax) cs[u] = fmaxf(cs[u], tl_row16_max(row >= 0 ? x : -INFINITY));
              else cs[u] += tl_row16_sum(row >= 0 ? x : 0.f);
            }
          }
        }
        if (d.pool) {
          if (l15 == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s_csum[wave * 16 + 4 * g4 + u] = cs[u];
          }
          __syncthreads();
          if (tid < 16) {
            float tot = s_pool[tid];
            for (int w = 0; w < TL_WAVES; ++w) tot = pmax ? fmaxf(tot, s_csum[w * 16 + tid]) : tot + s_csum[w * 16 + tid];
            s_pool[tid] = tot;
          }
        }
        if constexpr (TRACE) tr[4] += now() - t_r;
      }   // windows of the scan
      if (d.pool) {
        __syncthreads();
        if (tid < 16) {
          const float tot = s_pool[tid];
          if (d.pool == 1) {
            __hip_atomic_store(a.sums + (size_t)b * 128 + 16 * ct + tid, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            const int cntr = out_re - out_rs;
            float o;
            if (a.pool_mode == 2) o = cntr > 0 ? tot : 0.f;
            else {
              const float m = cntr > 0 ? tot / (float)cntr : 0.f;
              o = a.pool_mode == 1 ? powf(m, 1.f / a.gem_p[0]) : m;
            }
            a.out_global[(size_t)b * 256 + 16 * ct + tid] = o;
          }
        }
      }
    }   // column tiles
    if (!synced) cluster_wait();
    cluster_publish();
    trace_out();
  }
}

int tail_forward(const TailArgs& a, hipStream_t stream) {
  EGONN_REQUIRE(a.B >= 1 && a.flags && a.sums && a.err, EGONN_ERR_INVALID, "tail: bad arguments");
  static AttrOnce attr_done;
  if (attr_done.need()) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, TL_LDS));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tail_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, TL_LDS));
    attr_done.mark();
  }
  hipEvent_t* pev = prof_kernel_events();
  if (a.trace) {
    hipLaunchKernelGGL(tail_kernel<true>, dim3((unsigned)a.B * 8), dim3(TL_THREADS), TL_LDS, stream, a);
  } else if (pev[0]) {
    hipExtLaunchKernelGGL(tail_kernel<false>, dim3((unsigned)a.B * 8), dim3(TL_THREADS), TL_LDS, stream, pev[0], pev[1], 0, a);
    pev[0] = pev[1] = nullptr;
  } else {
    hipLaunchKernelGGL(tail_kernel<false>, dim3((unsigned)a.B * 8), dim3(TL_THREADS), TL_LDS, stream, a);
  }
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

}  // namespace egonn
