// Window-resident sparse convolution (k=3 maps of the big levels, fp32 feature maps) for gfx950.
//
// Replaces the same reference operators as sconv.hip / sconv_split.hip — MinkowskiConvolution(kernel_size=3) inside ME
// BasicBlock conv1 / conv2 (layers/eca_block.py:58-63, built at models/minkgl.py:121-134) — on the levels where a launch
// fills the chip:
//
//   out[o] = act( (sum_k in[nbr[o][k]] @ W[k]) * scale + shift )
//
// Why another kernel.  The gather kernels move every input row through the CU's vector-memory path once per OUTPUT
// NEIGHBOUR (6.3-7.4 times at levels 1-3, 9-10 with the padding of the 16-row groups) and — what turned out to cost as much
// as the matrix instructions themselves (tools/win_trace.py: 754 cycles per group-step against 193 of MFMA) — split it into
// bf16 hi + mid + lo every time (44 VALU per 16 rows x 32 channels; VALU and MFMA of the waves of a SIMD do not overlap).
// But a window of 256 consecutive Z-order rows references only 1.24-1.36 x 256 DISTINCT rows (tools/window_rows_study.py:
// mean 315-326): its own rows plus a thin halo.  So here
//   * a WORKGROUP owns one 256-row window of the row-group tables (16 groups; rowgroup.hip builds, per window, the list of
//     halo rows and the table entries as LDS slots);
//   * it loads the window's distinct rows ONCE (own rows: one contiguous range; halo rows: by index; full 128-byte lines),
//     splits them ONCE into bf16 hi / mid / lo and keeps the three planes in LDS in MFMA-operand order (one 32-channel
//     block at a time, 74 KB: two workgroups per CU);
//   * then every WAVE runs free, without any barrier: it owns G consecutive groups (64 rows for G = 4) and all output
//     columns, walks the union of its groups' offsets, takes the W fragments of a step (k, cb, 32-column slice) straight
//     from L2 into registers ONCE for its G groups (six 1 KB loads, double buffered), reads the three operand fragments
//     of a group from its rows' LDS slots (absent neighbour = the all-zero slot; three ds_read_b128, no arithmetic) and
//     issues the same 6-product sequence on v_mfma_f32_16x16x32_bf16 as the split kernel (fp32 accumulate, accumulators
//     never leave registers).  The step loop contains no vector arithmetic beyond addresses;
//   * with more than 32 input channels the window is re-staged per channel block (cb outer, k inner: two barriers per block).
// The vector-memory path then carries 1.3 x N rows + the W fragments + 2 bytes per table entry instead of 9-10 x N rows.
// Every output row is produced by one wave, summed in ascending channel block, ascending k, fixed term order: results do not
// depend on the batch, on the grouping of other rows, on which LDS slot a row landed in, or on eager vs graph execution.
// For CIN = 32 the order is the split kernel's (bitwise equal results).
// A halo row beyond a window's WIN_HALO slots (a few windows per batch; forced in the tests) keeps the table code
// WIN_OVF_SLOT: the lanes that meet it gather that operand from global memory and split it in registers — same values.
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
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

struct WinArgs {
  const float* in;           // [n_in][CIN] fp32
  const uint16_t* wslot;     // row-group tables, window-resident form (common.h RowGroups)
  const int32_t* urow;
  const int32_t* wmeta;
  const int32_t* snbr;       // global-row form (overflow windows only)
  const uint32_t* gmask;
  const int32_t* perm;
  const int32_t* meta;       // [0] = groups in use
  const void* Wsp;           // pack_split_weights (sconv_split.hip)
  const float* scale;        // folded BatchNorm (nullable)
  const float* shift;
  float* out;                // [n_out][COUT]
  float* psum;               // [groups][COUT] column sums of the stored values (nullable)
  uint32_t in_rows, w_bytes, tbl_bytes;
  int relu, cap_groups;
  unsigned long long* trace = nullptr;   // measurement builds only (tools/win_trace.py): 8 u64 per wave
  int abl = 0;                           // measurement builds only (WRONG results): 1 = every step reads the W fragments of
                                         // offset 0 (L1-resident), 2 = no MFMAs, 4 = operands always from the zero slot
};

__device__ static inline float win_row16_sum(float v) {   // sum over the 16 lanes of a DPP row (= the 16 rows of a tile)
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

// fp32 x 8 -> (hi, mid, lo) bf16 x 8, round to nearest even at every level (v_cvt_pk_bf16_f32); as sconv_split.hip
__device__ static inline void win_split8(const f32x4& a0, const f32x4& a1, bf16x8_t& hi, bf16x8_t& mid, bf16x8_t& lo) {
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float x0 = p < 2 ? a0[2 * p] : a1[2 * p - 4], x1 = p < 2 ? a0[2 * p + 1] : a1[2 * p - 3];
    const uint32_t hp = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){x0, x1}, bf16x2_t));
    const float r0 = x0 - __uint_as_float(hp << 16), r1 = x1 - __uint_as_float(hp & 0xFFFF0000u);
    const uint32_t mp = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){r0, r1}, bf16x2_t));
    const float s0 = r0 - __uint_as_float(mp << 16), s1 = r1 - __uint_as_float(mp & 0xFFFF0000u);
    h[p] = hp;
    m[p] = mp;
    l[p] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){s0, s1}, bf16x2_t));
  }
  hi = __builtin_bit_cast(bf16x8_t, (uint4){h[0], h[1], h[2], h[3]});
  mid = __builtin_bit_cast(bf16x8_t, (uint4){m[0], m[1], m[2], m[3]});
  lo = __builtin_bit_cast(bf16x8_t, (uint4){l[0], l[1], l[2], l[3]});
}

struct WinGeom {
  static constexpr int PLANE = (WIN_SLOTS + 1) * 64;          // one part (hi / mid / lo) of the staged rows of one 32-channel
                                                              // block: 64 bytes per slot, + the zero slot
  static constexpr int LDS_BYTES = 3 * PLANE;                 // 73 920: two workgroups per CU
};

// fp32 x 4 -> (hi, mid, lo) bf16 x 4 (the staging split: each element once per window)
__device__ static inline void win_split4(const f32x4& a, uint2& hi, uint2& mid, uint2& lo) {
  uint32_t h[2], m[2], l[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float x0 = a[2 * p], x1 = a[2 * p + 1];
    const uint32_t hp = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){x0, x1}, bf16x2_t));
    const float r0 = x0 - __uint_as_float(hp << 16), r1 = x1 - __uint_as_float(hp & 0xFFFF0000u);
    const uint32_t mp = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){r0, r1}, bf16x2_t));
    const float s0 = r0 - __uint_as_float(mp << 16), s1 = r1 - __uint_as_float(mp & 0xFFFF0000u);
    h[p] = hp;
    m[p] = mp;
    l[p] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){s0, s1}, bf16x2_t));
  }
  hi = make_uint2(h[0], h[1]);
  mid = make_uint2(m[0], m[1]);
  lo = make_uint2(l[0], l[1]);
}

// G: groups per wave (a workgroup has 16 / G waves).  The W fragments of a step are fetched once per wave: G = 4 moves a
// quarter of the W bytes of G = 1 through the vector-memory path.
template <int CIN, int COUT, int G, bool TRACE = false>
__global__ __launch_bounds__((16 / G) * 64, (CIN * COUT >= 64 * 64) ? 1 : 8 / G) void sconv_win_kernel(const WinArgs p) {   // (64 x 64: more
                                                                                      // than 256 registers rather than spills)
  auto now = [] { return (unsigned long long)__builtin_amdgcn_s_memtime(); };
  unsigned long long tr[8] = {};
  if constexpr (TRACE) tr[0] = now();
  constexpr int NW = 16 / G, NS = COUT / 32, NCB = CIN / 32, K = 27;
  constexpr int WSLICE = 6144;                           // bytes of one (k, cb, 32-column slice): six fragments
  constexpr int NOWN = WIN_ROWS / 8, NHALO = WIN_HALO / 8;   // staging pieces: 8 rows x 128 B per wave instruction
  constexpr int OPW = NOWN / NW, HPW = NHALO / NW;
  static_assert(NOWN % NW == 0 && NHALO % NW == 0, "pieces per wave");
  constexpr int PLANE = WinGeom::PLANE;
  extern __shared__ __attribute__((aligned(1024))) char smem[];   // [3 parts][WIN_SLOTS + 1][64 B]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l15 = lane & 15, g4 = lane >> 4;

  const int ngroups = __builtin_amdgcn_readfirstlane(min(p.meta[0], p.cap_groups));
  const int nwin = (ngroups + 15) >> 4;
  // contiguous eighth of the windows per XCD (block b runs on XCD b % 8): a Z-order slice of the map per L2
  const int cpx = (nwin + 7) >> 3;
  const int lt = blockIdx.x >> 3;
  const int w = (blockIdx.x & 7) * cpx + lt;
  if (lt >= cpx || w >= nwin) return;

  // ---- round trip 1: window header, halo rows, masks and output rows of this wave's groups
  const int4 wm = *reinterpret_cast<const int4*>(p.wmeta + (int64_t)w * 4);
  const int r0 = __builtin_amdgcn_readfirstlane(wm.x), rows = __builtin_amdgcn_readfirstlane(wm.y);
  const int nh = __builtin_amdgcn_readfirstlane(wm.z);
  const int novf = __builtin_amdgcn_readfirstlane(wm.w);   // halo rows without a slot (-1: hash set overflow)
  if (rows <= 0) return;
  int32_t hidx[HPW];
#pragma unroll
  for (int i = 0; i < HPW; ++i) hidx[i] = p.urow[(int64_t)w * WIN_HALO + (wave + NW * i) * 8 + (lane >> 3)];
  // this wave's groups: wave, wave + NW, wave + 2 NW, ... of the window's 16.  The window is sorted by neighbour mask, so
  // consecutive groups have the same (few or many) offsets: dealing them out round-robin gives every wave of the window the
  // same amount of work (consecutive quadruples measured 16 .. 101 group-steps per wave, and a window is as slow as its
  // slowest wave)
  const int gfirst = w * 16 + wave;
  auto gidx = [&](int j) { return gfirst + NW * j; };
  uint32_t gm[G];
  uint32_t U = 0;
  int32_t orow[G];
#pragma unroll
  for (int j = 0; j < G; ++j) {
    gm[j] = (gidx(j) < ngroups) ? p.gmask[gidx(j)] : 0u;
    gm[j] = __builtin_amdgcn_readfirstlane(gm[j]);
    U |= gm[j];
    orow[j] = (gm[j] >> 31) ? p.perm[(int64_t)gidx(j) * 16 + l15] : -1;
  }
  if (wave == 0 && lane < 12)                             // the zero slot of the three planes
    *reinterpret_cast<f32x4*>(smem + (lane >> 2) * PLANE + WIN_ZERO_SLOT * 64 + (lane & 3) * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};

  const __amdgpu_buffer_rsrc_t r_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in), 0, (int)(p.in_rows * (uint32_t)(CIN * 4)), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.Wsp), 0, (int)p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t t_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.wslot), 0, (int)p.tbl_bytes, 0x00020000);

  // ---- staging (round trip 2, and once more per further channel block): the window's distinct rows -> registers ->
  // bf16 hi / mid / lo -> LDS planes.  A wave instruction covers 8 rows x 128 B: lane L = row L >> 3, channels 4c..4c+3 with
  // c = L & 7.  The MFMA operand of lane group g holds channels {4g..4g+3, 16+4g..16+4g+3} (sp_chan of sconv_split.hip), so
  // chunk c lands in half (c >> 2) of the 16-byte operand (c & 3), XOR-swizzled by two slot bits against bank conflicts of
  // the random-row reads: byte (slot * 64) + (((c & 3) ^ ((slot >> 2) & 3)) * 16) + (c >> 2) * 8 of every plane.
  const int st_c = lane & 7;
  auto stage_pieces = [&](auto HALO, int cb) {              // own rows, then halo rows: two batches of loads (registers)
    constexpr bool halo = decltype(HALO)::value;
    constexpr int NP = halo ? HPW : OPW;
    f32x4 v[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      int32_t vi;
      if constexpr (halo) {
        vi = hidx[i];
      } else {
        const int slot = (wave + NW * i) * 8 + (lane >> 3);
        vi = slot < rows ? r0 + slot : -1;                                // -1: out of range -> zeros, no traffic
      }
      v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, vi * (CIN * 4) + st_c * 16, cb * 128, 0));
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int pc = halo ? NOWN + wave + NW * i : wave + NW * i;
      const bool live = halo ? (wave + NW * i) * 8 < nh : (wave + NW * i) * 8 < rows;       // wave-uniform
      if (!live) continue;
      const int slot = pc * 8 + (lane >> 3);
      uint2 hi, mid, lo;
      win_split4(v[i], hi, mid, lo);
      char* dst = smem + slot * 64 + (((st_c & 3) ^ ((slot >> 2) & 3)) * 16) + (st_c >> 2) * 8;
      *reinterpret_cast<uint2*>(dst) = hi;
      *reinterpret_cast<uint2*>(dst + PLANE) = mid;
      *reinterpret_cast<uint2*>(dst + 2 * PLANE) = lo;
    }
  };
  auto stage = [&](int cb) {
    stage_pieces(std::integral_constant<bool, false>{}, cb);
    stage_pieces(std::integral_constant<bool, true>{}, cb);
  };
  if constexpr (TRACE) tr[1] = now();

  // ---- per-lane constants of the operand reads
  const int w_lane = lane * 16;
  auto wload = [&](int k, int cb, int ns, f32x4 (&wf)[6]) {              // six fragments of slice ns of step (k, cb)
    const bool valid = k < 27;
    if constexpr (TRACE) { if ((p.abl & 1) && valid) k = 0; }
    const int so = valid ? ((k * NCB + cb) * NS + ns) * WSLICE : 0;
    const int vo = valid ? w_lane : (int)(0x80000000u | (uint32_t)w_lane);   // past the end: out of range, no traffic
#pragma unroll
    for (int f = 0; f < 6; ++f) wf[f] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, vo + f * 1024, so, 0));
  };
  auto mfma = [&](const f32x4& wf, const bf16x8_t& af, f32x4& c) {
    if constexpr (TRACE) { if (p.abl & 2) { c += wf * __builtin_bit_cast(f32x4, af)[0]; return; } }
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf), af, c, 0, 0, 0);
  };
  auto products = [&](const f32x4 (&wf)[6], const bf16x8_t (&a)[3], f32x4& c0, f32x4& c1) {
    // small terms first; wf[2 * part + nt], a[part] (0 hi, 1 mid, 2 lo)
    mfma(wf[4], a[0], c0); mfma(wf[5], a[0], c1);      // w lo * a hi
    mfma(wf[0], a[2], c0); mfma(wf[1], a[2], c1);      // w hi * a lo
    mfma(wf[2], a[1], c0); mfma(wf[3], a[1], c1);      // mid * mid
    mfma(wf[2], a[0], c0); mfma(wf[3], a[0], c1);      // w mid * a hi
    mfma(wf[0], a[1], c0); mfma(wf[1], a[1], c1);      // w hi * a mid
    mfma(wf[0], a[0], c0); mfma(wf[1], a[0], c1);      // hi * hi
  };

  f32x4 acc[G][NS][2];
#pragma unroll
  for (int j = 0; j < G; ++j)
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) acc[j][ns][0] = acc[j][ns][1] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const uint32_t mk_all = U & 0x07FFFFFFu;
  const int n_iter = (__popc(mk_all) + 3) >> 2;

  // One pass over the offsets of this wave's groups for channel block cb.  A global round trip costs 2-3 k cycles on a busy
  // chip (tools/win_trace.py: with one step of look-ahead the loop took 2 100 cycles per step whatever it computed), a step
  // is 4 group-steps of ~200 cycles of MFMA, and LDS leaves room for two waves per SIMD only — so everything that comes from
  // global memory is requested several steps ahead through register rings with static indices (the loop is unrolled by 4):
  //   slots (table entries, 2 bytes per lane and group): ring of 4 steps, requested 3 steps ahead;
  //   W fragments: ring of RB = 4 slices (NS = 1: requested 3 steps ahead; NS = 2: the two slices of the next step);
  //   operand fragments (LDS): ring of 2 groups — the three ds_read_b128 of the next group (or of the next step's first
  //   group) are issued before the MFMAs of the current one.
  // Slots and operands are read for all G groups (a group without the offset has the zero slot in its table); only the MFMAs
  // are skipped for it.
  constexpr int RB = 4, PD = RB / NS - 1;
  static_assert(NS == 1 || NS == 2, "W ring");
  static_assert(G % 2 == 0, "operand ring parity");
  auto pass = [&](auto OVF, int cb) {
    constexpr bool ovf_ = decltype(OVF)::value;           // the window has table entries without an LDS slot (rare)
    f32x4 wb[RB][6];
    bf16x8_t A[2][3];
    uint32_t sl[4][G];
    auto slots = [&](int k, uint32_t (&s_)[G]) {
#pragma unroll
      for (int j = 0; j < G; ++j) {
        if (k >= 27) s_[j] = WIN_ZERO_SLOT;                                // (scalar condition)
        else s_[j] = (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(t_rsrc, l15 * 2, (gidx(j) * K + k) * 32, 0);
      }
    };
    auto operand = [&](int k, int j, uint32_t slot, bf16x8_t (&a)[3]) {
      const bool ov = ovf_ && slot == (uint32_t)WIN_OVF_SLOT;
      uint32_t s = ov ? (uint32_t)WIN_ZERO_SLOT : slot;
      if constexpr (TRACE) { if (p.abl & 4) s = WIN_ZERO_SLOT; }
      const char* src = smem + s * 64 + ((((uint32_t)g4) ^ ((s >> 2) & 3u)) * 16);
      a[0] = *reinterpret_cast<const bf16x8_t*>(src);
      a[1] = *reinterpret_cast<const bf16x8_t*>(src + PLANE);
      a[2] = *reinterpret_cast<const bf16x8_t*>(src + 2 * PLANE);
      if constexpr (ovf_) if (__builtin_amdgcn_ballot_w64(ov)) {           // (wave-uniform, rare) rows without an LDS slot
        const int32_t r = ov ? p.snbr[((int64_t)gidx(j) * K + k) * 16 + l15] : -1;
        const int off = r * (CIN * 4) + cb * 128 + g4 * 16;                // row -1: beyond the buffer -> zeros
        const f32x4 x0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, off, 0, 0));
        const f32x4 x1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, off + 64, 0, 0));
        bf16x8_t h, m, l;
        win_split8(x0, x1, h, m, l);
        if (ov) { a[0] = h; a[1] = m; a[2] = l; }
      }
    };
    uint32_t mk = mk_all;
    auto gen = [&]() {
      const int k = mk ? __builtin_ctz(mk) : 27;
      mk &= mk - 1;
      return k;
    };
    int k0 = gen(), k1 = gen(), k2 = gen(), k3 = gen();
    slots(k0, sl[0]);
    slots(k1, sl[1]);
    slots(k2, sl[2]);
    {                                                                       // W of steps 0 .. PD-1
      const int kk[3] = {k0, k1, k2};
#pragma unroll
      for (int d = 0; d < PD; ++d)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns) wload(kk[d], cb, ns, wb[(d * NS + ns) % RB]);
    }
    operand(k0, 0, sl[0][0], A[0]);

    // Q = step index mod 4 (ring positions are static)
    auto step = [&](auto Q) {
      constexpr int q = decltype(Q)::value;
      const int kpd = PD == 1 ? k1 : (PD == 2 ? k2 : k3);
#pragma unroll
      for (int ns = 0; ns < NS; ++ns) wload(kpd, cb, ns, wb[((q + PD) * NS + ns) % RB]);
      slots(k3, sl[(q + 3) & 3]);
#pragma unroll
      for (int j = 0; j < G; ++j) {
        __builtin_amdgcn_sched_barrier(0);
        if (j + 1 < G) operand(k0, j + 1, sl[q][(j + 1) % G], A[(j + 1) & 1]);
        else operand(k1, 0, sl[(q + 1) & 3][0], A[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (k0 < 27 && ((gm[j] >> k0) & 1u)) {                            // wave-uniform
#pragma unroll
          for (int ns = 0; ns < NS; ++ns) products(wb[(q * NS + ns) % RB], A[j & 1], acc[j][ns][0], acc[j][ns][1]);
        }
      }
      k0 = k1; k1 = k2; k2 = k3;
      k3 = gen();
    };
#pragma unroll 1
    for (int it = 0; it < n_iter; ++it) {
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{});
    }
  };

#pragma unroll 1
  for (int cb = 0; cb < NCB; ++cb) {
    if (cb > 0) __syncthreads();                         // every wave is done with the planes of the previous channel block
    stage(cb);
    if constexpr (TRACE) { if (cb == 0) tr[2] = now(); }
    __syncthreads();
    if constexpr (TRACE) { if (cb == 0) tr[3] = now(); }
    if (novf != 0) pass(std::integral_constant<bool, true>{}, cb);
    else pass(std::integral_constant<bool, false>{}, cb);
  }
  if constexpr (TRACE) tr[4] = now();

  // ---- epilogue: BN scale/shift (+ReLU), one 16-byte store per tile; optional per-group column sums
#pragma unroll
  for (int j = 0; j < G; ++j) {
    if (!(gm[j] >> 31)) continue;
    const int32_t row = orow[j];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
      float sums[2][4];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int c0 = ns * 32 + nt * 16 + 4 * g4;
        f32x4 v = acc[j][ns][nt];
        if (p.scale) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + c0);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + c0);
          v = v * sc + sh;
        }
        if (p.relu) {
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = fmaxf(v[u], 0.f);
        }
        if (row >= 0) *reinterpret_cast<f32x4*>(p.out + (int64_t)row * COUT + c0) = v;
#pragma unroll
        for (int u = 0; u < 4; ++u) sums[nt][u] = row >= 0 ? v[u] : 0.f;
      }
      if (p.psum) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int u = 0; u < 4; ++u) sums[nt][u] = win_row16_sum(sums[nt][u]);
        if (l15 == 0) {
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
            *reinterpret_cast<f32x4*>(p.psum + (int64_t)gidx(j) * COUT + ns * 32 + nt * 16 + 4 * g4) =
                (f32x4){sums[nt][0], sums[nt][1], sums[nt][2], sums[nt][3]};
        }
      }
    }
  }
  if constexpr (TRACE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tr[5] = now();
    tr[6] = (unsigned long long)__popc(mk_all) | ((unsigned long long)(uint32_t)wm.w << 32) | ((unsigned long long)nh << 48);
    int present = 0;
#pragma unroll
    for (int j = 0; j < G; ++j) present += __popc(gm[j] & 0x07FFFFFFu);
    tr[7] = (unsigned long long)present;
    if (lane == 0 && p.trace) {
      unsigned long long* o = p.trace + ((int64_t)w * NW + wave) * 8;
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = tr[q];
    }
  }
}

template <int CIN, int COUT, int G, bool TRACE = false>
static int launch_win(const WinArgs& a, int64_t groups_hint, hipStream_t stream) {
  static AttrOnce attr_done;
  if (attr_done.need()) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&sconv_win_kernel<CIN, COUT, G, TRACE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, WinGeom::LDS_BYTES));
    attr_done.mark();
  }
  int64_t grid = std::max<int64_t>(cdiv(groups_hint, 16), 8);
  grid = (grid + 7) / 8 * 8;
  hipEvent_t* pev = prof_kernel_events();
  if (pev[0]) {      // bench.py roofline leg: time exactly this dispatch
    hipExtLaunchKernelGGL((sconv_win_kernel<CIN, COUT, G, TRACE>), dim3((unsigned)grid), dim3((16 / G) * 64), WinGeom::LDS_BYTES, stream,
                          pev[0], pev[1], 0, a);
    pev[0] = pev[1] = nullptr;
  } else {
    hipLaunchKernelGGL((sconv_win_kernel<CIN, COUT, G, TRACE>), dim3((unsigned)grid), dim3((16 / G) * 64), WinGeom::LDS_BYTES, stream, a);
  }
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// the channel plans of the k=3 convolutions of levels 1-3 (+ the input-gradient plan of the 32->64 layer)
bool sconv_win_supported(int cin, int cout) {
  static const int plans[][2] = {{32, 32}, {32, 64}, {64, 64}, {64, 32}};
  for (const auto& pl : plans)
    if (pl[0] == cin && pl[1] == cout) return true;
  return false;
}

// cfg: groups per wave (0 = product choice)
int sconv_win_forward(const float* in, int64_t n_in_cap, const RowGroups& rg, int64_t groups_hint, const void* Wsp, int cin,
                      int cout, const float* scale, const float* shift, int relu, float* out, float* psum, hipStream_t stream,
                      int cfg) {
  EGONN_REQUIRE(rg.built && rg.wslot && rg.win == WIN_ROWS && rg.K == 27, EGONN_ERR_STATE, "sconv(win): window tables not built");
  EGONN_REQUIRE(sconv_win_supported(cin, cout), EGONN_ERR_INVALID, "sconv(win): channel plan %d->%d not supported", cin, cout);
  EGONN_REQUIRE((uint64_t)n_in_cap * cin * 4 < (1ull << 32) - (1ull << 20), EGONN_ERR_INVALID,
                "sconv: input feature map of %lld rows exceeds the 4 GiB buffer-resource range", (long long)n_in_cap);
  if (groups_hint <= 0) return EGONN_OK;
  WinArgs a;
  a.in = in; a.wslot = rg.wslot; a.urow = rg.urow; a.wmeta = rg.wmeta; a.snbr = rg.snbr; a.gmask = rg.gmask; a.perm = rg.perm;
  a.meta = rg.meta; a.Wsp = Wsp; a.scale = scale; a.shift = shift; a.out = out; a.psum = psum;
  a.in_rows = (uint32_t)n_in_cap;
  a.w_bytes = (uint32_t)((uint64_t)rg.K * cin * cout * 6);
  a.tbl_bytes = (uint32_t)((uint64_t)rg.cap_groups * rg.K * 32);
  a.relu = relu ? 1 : 0; a.cap_groups = rg.cap_groups;
  static const int env_g = [] {                          // EGONN_WIN_G: measurement override
    const char* e = getenv("EGONN_WIN_G");
    return e ? atoi(e) : 0;
  }();
  int g = cfg ? cfg : (env_g ? env_g : 4);
  if (g >= 900) {      // 5900 + G: traced build (tools/win_trace.py)
    g -= 900;
    a.trace = g_sconv_trace;
    const char* e = getenv("EGONN_WIN_ABL");
    a.abl = e ? atoi(e) : 0;
    if (cin == 32 && cout == 32 && g == 4) return launch_win<32, 32, 4, true>(a, groups_hint, stream);
    if (cin == 64 && cout == 64 && g == 4) return launch_win<64, 64, 4, true>(a, groups_hint, stream);
  }
#define EGONN_WIN_CASE(CI, CO, GG) \
  if (cin == CI && cout == CO && g == GG) return launch_win<CI, CO, GG>(a, groups_hint, stream);
  EGONN_WIN_CASE(32, 32, 4) EGONN_WIN_CASE(32, 64, 4) EGONN_WIN_CASE(64, 64, 4) EGONN_WIN_CASE(64, 32, 4)
#undef EGONN_WIN_CASE
  set_error("sconv(win): no instantiation for %d->%d G %d", cin, cout, g);
  return EGONN_ERR_INVALID;
}

}  // namespace egonn
