// Row-group tables: the plan-time form of a kernel map that the sparse-convolution kernel (sconv.hip) consumes.
//
// A kernel map nbr[o][k] (k=3: 27 offsets, k=2/s=2 and its transpose: 8 slots; coords.hip) says which input row feeds
// output row o through weight slot k.  The convolution kernel keeps the accumulators of 16 output rows in MFMA
// registers and walks the offsets k; it can skip an offset only when NONE of its 16 rows has a neighbour there.  So
// the rows of a level are regrouped once per plan (shared by conv1/conv2 of a block, by every use in training and by
// both precisions):
//   * the rows of every sample are cut into windows of WIN consecutive rows (Z-order => spatially compact, and a
//     window never straddles two samples, so per-group column sums are per-sample sums: the ECA pooling of
//     layers/eca_block.py:21-36 falls out of the conv2 epilogue);
//   * inside a window the rows are sorted by their neighbour-presence bitmask (rare offsets most significant), so
//     that 16 consecutive sorted rows — a GROUP — have nearly the same set of present offsets.  Measured on the
//     benchmark clouds: MFMA work = 1.40-1.49 x the pair count (unsorted 16-row groups: 1.8-2.0 x, dense 27: 3.5-4.3 x);
//   * per group g:  gmask[g] = OR of the 16 masks (bit 31: group holds at least one real row),
//                   perm[g*16 + s] = output row of slot s (-1 = padding),
//                   snbr[(g*K + k)*16 + s] = input row (-1 = absent): one coalesced 64-byte line per (g, k).
// Window slot w owns the WIN/16 groups [w*GPW, (w+1)*GPW); partial windows leave trailing groups empty.  All sizes
// are read from device memory (row count, per-sample offsets), so the builder needs no host knowledge of N_l.
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace egonn {

// sort position of kernel offset k (27-offset maps): rare offsets (corners) most significant, the centre least.
// Class = number of non-zero components of the offset; inside a class, offsets with dz != 0 first (LiDAR surfaces
// are mostly horizontal or vertical sheets: measured presence 0.04 corners, 0.13-0.24 dz-edges/faces, 0.35-0.5
// in-plane, 1.0 centre).
__device__ static inline uint32_t remap27(uint32_t m) {
  // order[i] = offset index placed at bit i (LSB first): centre, in-plane faces, in-plane diagonals, dz faces,
  // dz edges, corners
  constexpr int order[27] = {13, 12, 14, 10, 16, 9, 11, 15, 17, 4, 22, 1, 7, 3, 5, 19, 25, 21, 23, 0, 2, 6, 8, 18, 20, 24, 26};
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 27; ++i) r |= ((m >> order[i]) & 1u) << i;
  return r;
}

struct RGJob {
  const int32_t* nbr;       // [n][K]
  const int32_t* n_dev;     // rows of the output level (device)
  const int32_t* boff;      // [B+1] per-sample row offsets of the output level (device)
  int32_t* perm;
  int32_t* snbr;
  uint32_t* gmask;
  int32_t* meta;            // [0] = number of groups, [1 + b] = first group of sample b (b = 0..B)
  int K, win, wbase;        // wbase: first block of this job in the launch
  int cap_rows, cap_groups; // capacity of the level / of the tables: a batch that exceeds a reservation is clipped, never
                            // written out of bounds (egonn_plan_status reports it)
};
struct RGArgs {
  RGJob job[RG_MAX_JOBS];
  int njobs, B, nblocks;
  int first_pass;              // 27-offset maps: radix passes first_pass .. 2 of the window sort (0 = all 27 mask bits)
  unsigned long long* trace;   // measurement hook (egonn_debug_set_trace): 8 s_memtime stamps per window; null = off
};

static constexpr int RG_THREADS = 512;                  // 8 waves per window; 55 KB table + 19 KB static LDS => two windows per CU
                                                        // (1024-row windows with 1024 threads took a whole CU each: 328 large windows on
                                                        // 256 CUs ran as two rounds, 46 us per step at batch 16, 187 at batch 64)
static constexpr int RG_WAVES = RG_THREADS / 64;
template <bool TRACE>    // TRACE: measurement build (tools/rowgroup_trace.py), s_memtime stamps per phase; the release build has none
__global__ __launch_bounds__(RG_THREADS) void rowgroup_build_kernel(RGArgs a) {
  // The window's slice of the kernel map (WIN x K ints, contiguous in memory: 55 KB for K = 27, WIN = 512) is read ONCE, fully
  // coalesced, into LDS; masks, sort and the transposed emit all work from there.  (The first version read the rows twice
  // from global memory in 108-byte pieces — the builder was bound by those partial-line requests, 66 us per step.)
  extern __shared__ __attribute__((aligned(16))) int32_t stbl[];          // [WIN][K]
  __shared__ unsigned long long skey[RG_MAX_WIN];
  __shared__ uint32_t smask[RG_MAX_WIN];
  __shared__ uint16_t s_cnt[RG_WAVES][512];              // radix sort: per-wave digit counts -> per-wave offsets (<= 1024)
  __shared__ uint32_t s_base[512];
  __shared__ uint32_t s_k[RG_MAX_WIN];
  __shared__ uint16_t s_i[RG_MAX_WIN];
  __shared__ int32_t s_info[4];
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned long long* const tr = (TRACE && a.trace) ? a.trace + (size_t)blockIdx.x * 8 : nullptr;
  auto stamp = [&](int i) {
    if constexpr (TRACE) {
      if (tr && tid == 0) tr[i] = __builtin_amdgcn_s_memtime();
    }
  };
  stamp(0);
  int j = 0;
  while (j + 1 < a.njobs && (int)blockIdx.x >= a.job[j + 1].wbase) ++j;
  const RGJob& J = a.job[j];
  const int w = blockIdx.x - J.wbase;
  const int K = J.K, WIN = J.win, GPW = WIN / 16, B = a.B;
  const int nlim = min(*J.n_dev, J.cap_rows);           // rows that exist: per-sample offsets are clipped to it
  auto boff_at = [&](int b) { return min(J.boff[b], nlim); };

  // ---- window -> (sample, first row); wave 0 scans the samples 64 at a time
  if (tid < 64) {
    int cum = 0, found_b = -1, found_first = 0;
    for (int b0 = 0; b0 < B; b0 += 64) {
      const int b = b0 + lane;
      const int nb = (b < B) ? (boff_at(b + 1) - boff_at(b)) : 0;
      const int nw = (nb + WIN - 1) / WIN;
      int incl = nw;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
      }
      const int first = cum + incl - nw;
      if (w == 0 && b < B) J.meta[1 + b] = min(first * GPW, J.cap_groups);
      if (b < B && w >= first && w < first + nw) { found_b = b; found_first = first; }
      cum += __shfl(incl, 63, 64);
    }
    if (w == 0 && lane == 0) {
      J.meta[0] = min(cum * GPW, J.cap_groups);
      J.meta[1 + B] = min(cum * GPW, J.cap_groups);
    }
    // exactly one lane (or none) found the window
    const unsigned long long m = __ballot(found_b >= 0);
    if (m) {
      const int src = __ffsll((long long)m) - 1;
      const int fb = __shfl(found_b, src, 64), ff = __shfl(found_first, src, 64);
      if (lane == 0) { s_info[0] = fb; s_info[1] = ff; }
    } else if (lane == 0) {
      s_info[0] = -1;
    }
  }
  __syncthreads();
  const int sb = s_info[0];
  if (sb < 0) return;                                   // window slot beyond the last sample
  const int r0 = boff_at(sb) + (w - s_info[1]) * WIN;
  const int rows = min(WIN, boff_at(sb + 1) - r0);

  stamp(1);
  // ---- the window's table -> LDS (coalesced dword loads; rows beyond the sample are never indexed)
  const int32_t* src = J.nbr + (int64_t)r0 * K;
  const int nint = rows * K;
  for (int i0 = 0; i0 < nint; i0 += 9 * RG_THREADS) {     // 9 independent loads per thread in flight (27 = 3 x 9)
    int32_t v[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const int i = i0 + u * RG_THREADS + tid;
      v[u] = (i < nint) ? src[i] : -1;
    }
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const int i = i0 + u * RG_THREADS + tid;
      if (i < nint) stbl[i] = v[u];
    }
  }
  __syncthreads();
  stamp(2);
  // ---- presence masks: one row per thread (row stride K = 27 or 8 dwords: 27 is odd => conflict-free)
  for (int i = tid; i < WIN; i += RG_THREADS) {
    uint32_t m = 0;
    if (i < rows) {
      const int32_t* row = stbl + i * K;
      if (K == 27) {
#pragma unroll
        for (int k = 0; k < 27; ++k) m |= (uint32_t)(row[k] >= 0) << k;
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) m |= (uint32_t)(row[k] >= 0) << k;
      }
    }
    smask[i] = m;
  }
  __syncthreads();
  stamp(3);
  // ---- stable LSD radix sort of the window by the (remapped) mask, 9 bits per pass, one element per thread; ties keep
  // the row order, padding (mask 0xFFFFFFFF) ends up last.  3 passes for the 27-offset maps, 1 for the 8-slot maps.
  // (A 55-stage bitonic network through LDS cost 25 800 cycles of a window's 47 600 — tools/rowgroup_trace.py.)
  static_assert(RG_THREADS == RG_MAX_WIN, "one element per thread");
  {
    const int wave = tid >> 6;
    const bool act = tid < WIN;
    uint32_t km = 0xFFFFFFFFu;
    uint32_t idx = (uint32_t)tid;
    if (tid < rows) km = (K == 27) ? remap27(smask[tid]) : smask[tid];
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int npass = (K == 27) ? 3 : 1;
    for (int pass = (K == 27 ? a.first_pass : 0); pass < npass; ++pass) {
      for (int e = tid; e < RG_WAVES * 512 / 2; e += RG_THREADS) reinterpret_cast<uint32_t*>(&s_cnt[0][0])[e] = 0u;
      __syncthreads();
      const uint32_t d = (km >> (9 * pass)) & 511u;
      unsigned long long peers = __ballot(act);
#pragma unroll
      for (int bit = 0; bit < 9; ++bit) {
        const bool on = (d >> bit) & 1u;
        const unsigned long long bal = __ballot(on);
        peers &= on ? bal : ~bal;
      }
      const int rank = __popcll(peers & lt), cnt = __popcll(peers);
      if (act && rank == 0) s_cnt[wave][d] = (uint16_t)cnt;
      __syncthreads();
      if (tid < 512) {                                   // digit tid: exclusive prefix over the waves, total of the digit
        uint32_t run = 0;
#pragma unroll
        for (int wv = 0; wv < RG_WAVES; ++wv) {
          const uint32_t c = s_cnt[wv][tid];
          s_cnt[wv][tid] = (uint16_t)run;
          run += c;
        }
        s_base[tid] = run;
      }
      __syncthreads();
      if (tid < 64) {                                    // exclusive scan of the 512 digit totals: 8 per lane
        uint32_t v[8], sum = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) { v[q] = s_base[8 * lane + q]; sum += v[q]; }
        uint32_t incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t up = __shfl_up(incl, o, 64);
          if (lane >= o) incl += up;
        }
        uint32_t run = incl - sum;
#pragma unroll
        for (int q = 0; q < 8; ++q) { s_base[8 * lane + q] = run; run += v[q]; }
      }
      __syncthreads();
      if (act) {
        const uint32_t pos = s_base[d] + s_cnt[wave][d] + (uint32_t)rank;
        s_k[pos] = km;
        s_i[pos] = (uint16_t)idx;
      }
      __syncthreads();
      if (act) { km = s_k[tid]; idx = s_i[tid]; }
    }
    if (act) skey[tid] = (km != 0xFFFFFFFFu) ? (((unsigned long long)km << 16) | idx) : ~0ull;
  }
  __syncthreads();
  stamp(4);
  // ---- perm + group masks
  const int64_t gbase = (int64_t)w * GPW;
  for (int i = tid; i < WIN; i += RG_THREADS) {         // WIN is a multiple of 64: whole waves stay in the loop
    const unsigned long long key = skey[i];
    const bool valid = key != ~0ull;
    const int lr = (int)(key & 0xFFFFu);
    J.perm[gbase * 16 + i] = valid ? r0 + lr : -1;
    uint32_t m = valid ? (smask[lr] | 0x80000000u) : 0u;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) m |= __shfl_xor(m, o, 64);
    if ((i & 15) == 0) J.gmask[gbase + (i >> 4)] = m;
  }
  stamp(5);
  // ---- sorted table, one 64-byte line per (group, offset): thread = one 16-byte piece (4 consecutive slots of one
  // offset), its four values gathered from the LDS table (rows are random, the column is fixed: stride-27 rows spread
  // over the banks), written as the group's contiguous K x 64-byte block
  const int ngr = (rows + 15) >> 4;                     // groups with real rows
  const int ppg = K * 4;                                // pieces per group
  for (int e = tid; e < ngr * ppg; e += RG_THREADS) {
    const int gl = e / ppg, pc = e - gl * ppg;
    const int k = pc >> 2, s0 = (pc & 3) * 4;
    int32_t v[4];
#pragma unroll
    for (int j2 = 0; j2 < 4; ++j2) {
      const unsigned long long key = skey[gl * 16 + s0 + j2];
      v[j2] = (key != ~0ull) ? stbl[(int)(key & 0xFFFFu) * K + k] : -1;
    }
    reinterpret_cast<int4*>(J.snbr + (gbase + gl) * K * 16)[pc] = make_int4(v[0], v[1], v[2], v[3]);
  }
  stamp(6);
  if constexpr (TRACE) {
    if (tr && tid == 0) tr[7] = (unsigned long long)K | ((unsigned long long)rows << 8);
  }
}

// Dispatch order of the split-bf16 kernel's tasks (4 consecutive groups each; sconv_split.hip).  A launch is a few rounds
// of tasks whose length (offsets present in the union of the task's groups) varies 5..25 steps: in table order the last
// round is as long as its longest task and the chip idles behind it (tools/split_trace.py: 53 % packing on L1 k3).  Inside
// every contiguous eighth of the tasks (one XCD: the locality of the Z-order slice is kept) the tasks are therefore stably
// sorted by descending step count — longest first.  One workgroup of 8 waves per (map, eighth), every wave a contiguous slice:
// counting sort with ballot ranks.
struct RGOrderArgs {
  const uint32_t* gmask[RG_MAX_JOBS];
  const int32_t* meta[RG_MAX_JOBS];
  int32_t* order[RG_MAX_JOBS];
  int cap_groups[RG_MAX_JOBS];
  int njobs;
};
static constexpr int ORD_WAVES = 8;
__global__ __launch_bounds__(ORD_WAVES * 64) void rowgroup_order_kernel(const RGOrderArgs a) {
  const int j = blockIdx.x >> 3, xcd = blockIdx.x & 7, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (j >= a.njobs || !a.order[j]) return;
  const int ngroups = min(a.meta[j][0], a.cap_groups[j]);
  const int ntask = (ngroups + 3) >> 2;
  const int cpx = (ntask + 7) >> 3;
  const int t0 = xcd * cpx, t1 = min(t0 + cpx, ntask);
  // every wave owns a contiguous slice of the eighth (stable: slices in wave order)
  const int per_wave = ((t1 - t0 + ORD_WAVES - 1) / ORD_WAVES + 63) / 64 * 64;
  const int w0 = min(t0 + wave * per_wave, t1), w1 = min(w0 + per_wave, t1);
  const uint32_t* gm = a.gmask[j];
  auto cost_of = [&](int t) {
    const uint4 m = (4 * t + 3 < ngroups) ? *reinterpret_cast<const uint4*>(gm + 4 * t)
                                          : make_uint4(4 * t < ngroups ? gm[4 * t] : 0u, 4 * t + 1 < ngroups ? gm[4 * t + 1] : 0u,
                                                       4 * t + 2 < ngroups ? gm[4 * t + 2] : 0u, 0u);
    const uint32_t u = m.x | m.y | m.z | m.w;
    return (u >> 31) ? __popc(u & 0x07FFFFFFu) : 0;      // a task of padding groups only costs nothing
  };
  // lanes of the same cost class among the 64 tasks of a batch: five ballots on the bits of the cost (0..27)
  auto peers_of = [&](int c, bool act) {
    unsigned long long p = __ballot(act);
#pragma unroll
    for (int bit = 0; bit < 5; ++bit) {
      const bool on = (c >> bit) & 1;
      const unsigned long long bal = __ballot(on);
      p &= on ? bal : ~bal;
    }
    return act ? p : 0ull;
  };
  __shared__ int32_t cnt[ORD_WAVES][32];
  if (lane < 32) cnt[wave][lane] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const unsigned long long lt = (1ull << lane) - 1ull;
  // pass 1: this wave's tasks per cost class (the first lane of every class adds its batch's count)
  for (int tb = w0; tb < w1; tb += 64) {
    const int t = tb + lane;
    const bool act = t < w1;
    const int c = act ? cost_of(t) : 0;
    const unsigned long long p = peers_of(c, act);
    if (act && (p & lt) == 0) cnt[wave][c] += __popcll(p);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __syncthreads();
  // descending order: class 27 first.  base[w][c] = tasks of classes > c (all waves) + tasks of class c in earlier waves
  if (wave == 0) {
    int32_t mine[ORD_WAVES], tot = 0;
#pragma unroll
    for (int w = 0; w < ORD_WAVES; ++w) { mine[w] = lane < 32 ? cnt[w][lane] : 0; tot += mine[w]; }
    int32_t start = 0, run = 0;
    for (int cc = 27; cc >= 0; --cc) {
      const int32_t n = __shfl(tot, cc, 64);
      if (lane == cc) start = run;
      run += n;
    }
    if (lane < 32) {
#pragma unroll
      for (int w = 0; w < ORD_WAVES; ++w) { cnt[w][lane] = start; start += mine[w]; }
    }
  }
  __syncthreads();
  // pass 2: stable scatter of this wave's slice
  for (int tb = w0; tb < w1; tb += 64) {
    const int t = tb + lane;
    const bool act = t < w1;
    const int c = act ? cost_of(t) : 0;
    const unsigned long long p = peers_of(c, act);
    const int32_t b = act ? cnt[wave][c] : 0;
    if (act) a.order[j][t0 + b + __popcll(p & lt)] = t;
    __builtin_amdgcn_wave_barrier();
    if (act && (p & lt) == 0) cnt[wave][c] = b + __popcll(p);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// Builds the row-group tables of `jobs` (all in one launch).  Every job's arrays must hold cap_groups groups.
int rowgroup_build(const RGBuild* jobs, int njobs, int B, hipStream_t stream) {
  EGONN_REQUIRE(njobs >= 1 && njobs <= RG_MAX_JOBS, EGONN_ERR_INVALID, "rowgroup_build: %d jobs", njobs);
  RGArgs a;
  a.njobs = njobs;
  a.B = B;
  int nb = 0;
  for (int j = 0; j < njobs; ++j) {
    const RGBuild& b = jobs[j];
    EGONN_REQUIRE(b.rg && b.nbr && b.n_dev && b.boff, EGONN_ERR_INVALID, "rowgroup_build: null job field");
    EGONN_REQUIRE(b.rg->win == 256 || b.rg->win == 512, EGONN_ERR_INVALID, "rowgroup window %d", b.rg->win);
    RGJob& J = a.job[j];
    J.nbr = b.nbr; J.n_dev = b.n_dev; J.boff = b.boff;
    J.perm = b.rg->perm; J.snbr = b.rg->snbr; J.gmask = b.rg->gmask; J.meta = b.rg->meta;
    J.K = b.rg->K; J.win = b.rg->win; J.wbase = nb;
    J.cap_rows = b.cap_rows > 0 ? b.cap_rows : INT32_MAX;
    J.cap_groups = b.rg->cap_groups;
    nb += b.rg->cap_groups / (b.rg->win / 16);
  }
  a.nblocks = nb;
  static const int env_first = [] { const char* e = getenv("EGONN_RG_FIRST_PASS"); return e ? atoi(e) : 0; }();   // measurement switch
  a.first_pass = std::min(std::max(env_first, 0), 2);
  a.trace = g_sconv_trace;
  if (nb == 0) return EGONN_OK;
  static AttrOnce attr_done;
  if (attr_done.need()) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgroup_build_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));   // + 19 KB static
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgroup_build_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    attr_done.mark(); 
  }
  size_t lds = 0;                                        // the largest window table of the launch
  for (int j = 0; j < njobs; ++j) lds = std::max(lds, (size_t)jobs[j].rg->win * jobs[j].rg->K * sizeof(int32_t));
  if (a.trace) hipLaunchKernelGGL(rowgroup_build_kernel<true>, dim3((unsigned)nb), dim3(RG_THREADS), lds, stream, a);
  else hipLaunchKernelGGL(rowgroup_build_kernel<false>, dim3((unsigned)nb), dim3(RG_THREADS), lds, stream, a);
  HIP_CHECK(hipGetLastError());
  RGOrderArgs oa;
  oa.njobs = njobs;
  bool any = false;
  for (int j = 0; j < njobs; ++j) {
    oa.gmask[j] = jobs[j].rg->gmask; oa.meta[j] = jobs[j].rg->meta; oa.order[j] = jobs[j].rg->order4;
    oa.cap_groups[j] = jobs[j].rg->cap_groups;
    any |= jobs[j].rg->order4 != nullptr;
  }
  if (any) {
    hipLaunchKernelGGL(rowgroup_order_kernel, dim3((unsigned)(njobs * 8)), dim3(ORD_WAVES * 64), 0, stream, oa);
    HIP_CHECK(hipGetLastError());
  }
  return EGONN_OK;
}

}  // namespace egonn
