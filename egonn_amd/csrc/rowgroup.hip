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
};
struct RGArgs {
  RGJob job[RG_MAX_JOBS];
  int njobs, B, nblocks;
};

static constexpr int RG_THREADS = 1024;                 // 16 waves per window: the per-group passes are latency chains
static constexpr int RG_WAVES = RG_THREADS / 64;
__global__ __launch_bounds__(RG_THREADS) void rowgroup_build_kernel(RGArgs a) {
  __shared__ unsigned long long skey[RG_MAX_WIN];
  __shared__ uint32_t smask[RG_MAX_WIN];
  __shared__ __attribute__((aligned(16))) int32_t stile[RG_WAVES * 27 * 16];
  __shared__ int32_t s_info[4];
  const int tid = threadIdx.x, lane = tid & 63;
  int j = 0;
  while (j + 1 < a.njobs && (int)blockIdx.x >= a.job[j + 1].wbase) ++j;
  const RGJob& J = a.job[j];
  const int w = blockIdx.x - J.wbase;
  const int K = J.K, WIN = J.win, GPW = WIN / 16, B = a.B;

  // ---- window -> (sample, first row); wave 0 scans the samples 64 at a time
  if (tid < 64) {
    int cum = 0, found_b = -1, found_first = 0;
    for (int b0 = 0; b0 < B; b0 += 64) {
      const int b = b0 + lane;
      const int nb = (b < B) ? (J.boff[b + 1] - J.boff[b]) : 0;
      const int nw = (nb + WIN - 1) / WIN;
      int incl = nw;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
      }
      const int first = cum + incl - nw;
      if (w == 0 && b < B) J.meta[1 + b] = first * GPW;
      if (b < B && w >= first && w < first + nw) { found_b = b; found_first = first; }
      cum += __shfl(incl, 63, 64);
    }
    if (w == 0 && lane == 0) {
      J.meta[0] = cum * GPW;
      J.meta[1 + B] = cum * GPW;
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
  const int r0 = J.boff[sb] + (w - s_info[1]) * WIN;
  const int rows = min(WIN, J.boff[sb + 1] - r0);

  // ---- presence masks.  A wave reads whole table rows with consecutive lanes (K = 27: 2 rows per load, lanes 0-26 and
  // 32-58; K = 8: 8 rows per load) — each load touches 2-4 cache lines instead of 64 — and a ballot IS the mask.
  const int32_t* src = J.nbr + (int64_t)r0 * K;
  const int wave = tid >> 6;
  // (8 loads are issued before the first ballot: a dependent load -> ballot -> LDS chain per row was latency bound)
  if (K == 27) {
    const int half = lane >> 5, kk = lane & 31;
    for (int rb = wave * 16; rb < WIN; rb += 16 * RG_WAVES) {   // waves x 8 loads x 2 rows
      int32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + 2 * u + half;
        v[u] = (r < rows && kk < 27) ? src[r * 27 + kk] : -1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + 2 * u + half;
        const unsigned long long bal = __ballot(v[u] >= 0);
        if (kk == 0 && r < WIN) smask[r] = (uint32_t)(bal >> (32 * half)) & 0x07FFFFFFu;
      }
    }
  } else {
    const int sub = lane >> 3, kk = lane & 7;
    for (int rb = wave * 64; rb < WIN; rb += 64 * RG_WAVES) {   // waves x 8 loads x 8 rows
      int32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + 8 * u + sub;
        v[u] = (r < rows) ? src[r * 8 + kk] : -1;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + 8 * u + sub;
        const unsigned long long bal = __ballot(v[u] >= 0);
        if (kk == 0 && r < WIN) smask[r] = (uint32_t)(bal >> (8 * sub)) & 0xFFu;
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < WIN; i += RG_THREADS) {
    unsigned long long key = ~0ull;
    if (i < rows) key = ((unsigned long long)(K == 27 ? remap27(smask[i]) : smask[i]) << 16) | (unsigned)i;
    skey[i] = key;
  }
  __syncthreads();
  // ---- bitonic sort of the window (ascending; padding keys are all-ones and end up last)
  for (int k2 = 2; k2 <= WIN; k2 <<= 1) {
    for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
      for (int e = tid; e < WIN / 2; e += RG_THREADS) {
        const int i = ((e & ~(j2 - 1)) << 1) | (e & (j2 - 1));
        const int p = i | j2;
        const bool up = (i & k2) == 0;
        const unsigned long long x = skey[i], y = skey[p];
        if ((x > y) == up) { skey[i] = y; skey[p] = x; }
      }
      __syncthreads();
    }
  }
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
  // ---- sorted table, one 64-byte line per (group, offset).  Per group a wave reads its 16 rows as whole rows (as
  // above), transposes them through a wave-private LDS tile [k][slot] and writes the group's contiguous K x 64-byte
  // block with 16-byte stores.
  const int ngr = (rows + 15) >> 4;                     // groups with real rows
  int32_t* tile = reinterpret_cast<int32_t*>(stile) + wave * (27 * 16);
  for (int gl = wave; gl < ngr; gl += RG_WAVES) {
    if (K == 27) {
      const int half = lane >> 5, kk = lane & 31;
      unsigned long long key[8];
      int32_t v[8];
#pragma unroll
      for (int h = 0; h < 8; ++h) key[h] = skey[gl * 16 + 2 * h + half];
#pragma unroll
      for (int h = 0; h < 8; ++h) v[h] = (kk < 27 && key[h] != ~0ull) ? src[(int)(key[h] & 0xFFFFu) * 27 + kk] : -1;
#pragma unroll
      for (int h = 0; h < 8; ++h)
        if (kk < 27) tile[kk * 16 + 2 * h + half] = v[h];
    } else {
      const int sub = lane >> 3, kk = lane & 7;
      unsigned long long key[2];
      int32_t v[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) key[h] = skey[gl * 16 + 8 * h + sub];
#pragma unroll
      for (int h = 0; h < 2; ++h) v[h] = (key[h] != ~0ull) ? src[(int)(key[h] & 0xFFFFu) * 8 + kk] : -1;
#pragma unroll
      for (int h = 0; h < 2; ++h) tile[kk * 16 + 8 * h + sub] = v[h];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int4* dst = reinterpret_cast<int4*>(J.snbr + (gbase + gl) * K * 16);
    const int n16 = K * 4;
    for (int e = lane; e < n16; e += 64) dst[e] = reinterpret_cast<const int4*>(tile)[e];
    __builtin_amdgcn_wave_barrier();
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
    EGONN_REQUIRE(b.rg->win == 256 || b.rg->win == 512 || b.rg->win == 1024, EGONN_ERR_INVALID, "rowgroup window %d", b.rg->win);
    RGJob& J = a.job[j];
    J.nbr = b.nbr; J.n_dev = b.n_dev; J.boff = b.boff;
    J.perm = b.rg->perm; J.snbr = b.rg->snbr; J.gmask = b.rg->gmask; J.meta = b.rg->meta;
    J.K = b.rg->K; J.win = b.rg->win; J.wbase = nb;
    nb += b.rg->cap_groups / (b.rg->win / 16);
  }
  a.nblocks = nb;
  if (nb == 0) return EGONN_OK;
  hipLaunchKernelGGL(rowgroup_build_kernel, dim3((unsigned)nb), dim3(RG_THREADS), 0, stream, a);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

}  // namespace egonn
