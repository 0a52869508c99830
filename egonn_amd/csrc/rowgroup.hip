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
//                   snbr[(g*K + k)*16 + s] = input row + 1 (0 = absent): one coalesced 64-byte line per (g, k).
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

__global__ __launch_bounds__(256) void rowgroup_build_kernel(RGArgs a) {
  __shared__ unsigned long long skey[RG_MAX_WIN];
  __shared__ uint32_t smask[RG_MAX_WIN];
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

  // ---- presence masks: one thread per row, the K table entries of a row are independent loads (unrolled)
  const int32_t* src = J.nbr + (int64_t)r0 * K;
  auto row_mask = [&](const int32_t* rp, auto KK) {
    constexpr int kk = decltype(KK)::value;
    int32_t v[kk];
#pragma unroll
    for (int k = 0; k < kk; ++k) v[k] = rp[k];
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < kk; ++k) m |= (v[k] >= 0 ? 1u : 0u) << k;
    return m;
  };
  for (int i = tid; i < WIN; i += 256) {
    uint32_t m = 0;
    if (i < rows) m = (K == 27) ? row_mask(src + i * 27, std::integral_constant<int, 27>{}) : row_mask(src + i * 8, std::integral_constant<int, 8>{});
    smask[i] = m;
    unsigned long long key = ~0ull;
    if (i < rows) key = ((unsigned long long)(K == 27 ? remap27(m) : m) << 16) | (unsigned)i;
    skey[i] = key;
  }
  __syncthreads();
  // ---- bitonic sort of the window (ascending; padding keys are all-ones and end up last)
  for (int k2 = 2; k2 <= WIN; k2 <<= 1) {
    for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
      for (int e = tid; e < WIN / 2; e += 256) {
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
  for (int i = tid; i < WIN; i += 256) {                // WIN is a multiple of 256: all lanes stay in the loop
    const unsigned long long key = skey[i];
    const bool valid = key != ~0ull;
    const int lr = (int)(key & 0xFFFFu);
    J.perm[gbase * 16 + i] = valid ? r0 + lr : -1;
    uint32_t m = valid ? (smask[lr] | 0x80000000u) : 0u;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) m |= __shfl_xor(m, o, 64);
    if ((i & 15) == 0) J.gmask[gbase + (i >> 4)] = m;
  }
  // ---- sorted table, one 64-byte line per (group, offset): thread = sorted slot, K independent loads then K stores
  const int ngr = (rows + 15) >> 4;                     // groups with real rows
  auto emit = [&](int i, auto KK) {
    constexpr int kk = decltype(KK)::value;
    const unsigned long long key = skey[i];
    const bool valid = key != ~0ull;
    const int32_t* rp = src + (int)(key & 0xFFFFu) * kk;
    int32_t v[kk];
#pragma unroll
    for (int k = 0; k < kk; ++k) v[k] = valid ? rp[k] + 1 : 0;
    int32_t* dst = J.snbr + (gbase + (i >> 4)) * kk * 16 + (i & 15);
#pragma unroll
    for (int k = 0; k < kk; ++k) dst[k * 16] = v[k];
  };
  for (int i = tid; i < ngr * 16; i += 256) {
    if (K == 27) emit(i, std::integral_constant<int, 27>{});
    else emit(i, std::integral_constant<int, 8>{});
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
  hipLaunchKernelGGL(rowgroup_build_kernel, dim3((unsigned)nb), dim3(256), 0, stream, a);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

}  // namespace egonn
