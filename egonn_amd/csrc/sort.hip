// Stable LSD radix sort of (u64 key, u32 value) pairs, 8 bits per pass, hand-written for wave64.
//
// Per pass two launches:
//   hist    : per-tile digit histogram (LDS atomics) -> tilehist[tile][256]; integer global atomics
//             additionally accumulate per-supertile (32 tiles) and grand totals, so that the scatter
//             kernel can derive its global base offsets without a separate scan launch.
//   scatter : every wave ranks its keys with a ballot-based match-any (8 ballots per 64 keys),
//             wave-private LDS counters give the stable in-tile rank, then keys/values are written
//             to  digit_base + tiles_before + rank.
// Order inside a tile is (wave, round, lane) and waves own contiguous sub-ranges, so the sort is stable.
//
// HBM traffic per pass: 12 B read (hist: 8) + 12 B written per pair; the sort is launch/latency bound at
// the sizes of this path (<= a few M keys), not bandwidth bound.
#include "common.h"

namespace egonn {

static constexpr int SORT_BLOCK = 256;
static constexpr int SORT_WAVES = SORT_BLOCK / 64;
static constexpr int SORT_ROUNDS = 8;                        // keys per thread
static constexpr int SORT_TILE = SORT_BLOCK * SORT_ROUNDS;   // 2048 keys per workgroup (391 tiles for the 800 k points of a batch; 4096 left CUs idle, 1024 is no better)
static constexpr int SORT_SUPER = 32;                        // tiles per supertile

// n_dev (nullable): the true element count lives in device memory (capturable plans); n is then the capacity the grid
// was sized for.
__device__ static inline int64_t sort_count(int64_t n, const int64_t* __restrict__ n_dev) {
  if (!n_dev) return n;
  const int64_t v = *n_dev;
  return v < n ? v : n;
}
__global__ __launch_bounds__(SORT_BLOCK) void sort_hist_kernel(const uint64_t* __restrict__ keys, int64_t n_cap,
                                                               const int64_t* __restrict__ n_dev,
                                                               int shift, int32_t* __restrict__ tilehist,
                                                               int32_t* __restrict__ superhist,
                                                               int32_t* __restrict__ total) {
  __shared__ int32_t hist[256];
  const int tid = threadIdx.x;
  const int64_t n = sort_count(n_cap, n_dev);
  hist[tid] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * SORT_TILE;
  // wave-aggregated counting: the lanes of a wave that hold the same digit are found with a ballot match-any and ONE of them
  // adds their number — the high digits of a Z-order key are nearly constant inside a tile (the batch digit exactly), and
  // 2048 LDS atomics on one bin serialise (the histogram of such a pass took as long as the scatter)
  const int lane = tid & 63;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  uint64_t kreg[SORT_ROUNDS];
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r) {
    const int64_t i = base + (int64_t)r * SORT_BLOCK + tid;
    kreg[r] = (i < n) ? keys[i] : 0ull;
  }
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r) {
    const int64_t i = base + (int64_t)r * SORT_BLOCK + tid;
    const bool valid = i < n;
    const uint32_t d = (uint32_t)(kreg[r] >> shift) & 0xFF;
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1;
      const uint64_t bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    if (valid && (m & lt) == 0) atomicAdd(&hist[d], (int32_t)__popcll(m));
  }
  __syncthreads();
  const int32_t c = hist[tid];
  tilehist[(int64_t)blockIdx.x * 256 + tid] = c;
  // (no grand-total atomics: every tile of a pass adding to the same 256 words serialised — 1 563 tiles at batch 64 cost
  // more than reading the keys; the scatter sums the <= 64 supertile rows instead)
  if (c) atomicAdd(&superhist[(blockIdx.x / SORT_SUPER) * 256 + tid], c);
}

__global__ __launch_bounds__(SORT_BLOCK) void sort_scatter_kernel(
    const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint64_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, int64_t n_cap, const int64_t* __restrict__ n_dev, int shift,
    const int32_t* __restrict__ tilehist, const int32_t* __restrict__ superhist, const int32_t* __restrict__ total) {
  const int64_t n = sort_count(n_cap, n_dev);
  __shared__ int32_t whist[SORT_WAVES][256];   // running per-wave digit counts
  __shared__ int32_t dbase[256];               // global base of every digit for this tile
  __shared__ int32_t wsum[SORT_WAVES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int tile = blockIdx.x;
  // the tile's keys are requested first: they do not depend on the histogram sums below, whose L2 round trips they overlap
  const int64_t wbase = (int64_t)tile * SORT_TILE + (int64_t)wave * (64 * SORT_ROUNDS);
  uint64_t k[SORT_ROUNDS];
  uint32_t v[SORT_ROUNDS];
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r) {
    const int64_t i = wbase + r * 64 + lane;
    const bool valid = i < n;
    k[r] = valid ? keys_in[i] : ~0ull;
    v[r] = valid ? vals_in[i] : 0u;
  }

  // ---- global base for digit d = tid: (#keys with smaller digit) + (#keys with digit d in earlier tiles)
  {
    // (the loads of these two sums are issued 8 at a time: as a plain loop every load waited for the previous one, and ~30
    // L2 round trips were most of a tile's 15 us)
    int32_t before = 0, tot = 0;
    const int super = tile / SORT_SUPER;
    const int nsuper = (gridDim.x + SORT_SUPER - 1) / SORT_SUPER;
    for (int s0 = 0; s0 < nsuper; s0 += 8) {
      int32_t c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = (s0 + u < nsuper) ? superhist[(s0 + u) * 256 + tid] : 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        tot += c[u];
        if (s0 + u < super) before += c[u];
      }
    }
    for (int t0 = super * SORT_SUPER; t0 < tile; t0 += 8) {
      int32_t c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = (t0 + u < tile) ? tilehist[(int64_t)(t0 + u) * 256 + tid] : 0;
#pragma unroll
      for (int u = 0; u < 8; ++u) before += c[u];
    }
    // exclusive scan of the digit totals over the 256 digits (wave scan + 4 wave sums)
    int32_t incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int32_t v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) whist[w][tid] = 0;
    __syncthreads();
    int32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    dbase[tid] = woff + incl - tot + before;
  }
  __syncthreads();

  // ---- stable ranking: wave `wave` owns keys [base + wave*64*ROUNDS, +64*ROUNDS)
  int32_t rank[SORT_ROUNDS];
  volatile int32_t* my = whist[wave];
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r) {
    const int64_t i = wbase + r * 64 + lane;
    const bool valid = i < n;
    const uint32_t d = (uint32_t)(k[r] >> shift) & 0xFF;
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1;
      const uint64_t bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    const int32_t prior = my[d];
    __builtin_amdgcn_wave_barrier();
    rank[r] = prior + __popcll(m & lt);
    if (valid && (m & lt) == 0) my[d] = prior + __popcll(m);   // group leader publishes the new count
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // ---- thread = digit: the tile's count of the digit, the start of the digit's run INSIDE the tile (exclusive scan over
  // the digits), and per wave the local position of its first key of that digit
  __shared__ uint64_t lkey[SORT_TILE];          // the tile, locally sorted by digit (stable)
  __shared__ uint32_t lval[SORT_TILE];
  __shared__ int32_t delta[256];                // global position - local position, per digit
  {
    int32_t cnt = 0;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) cnt += whist[w][tid];
    int32_t incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int32_t u = __shfl_up(incl, o, 64);
      if (lane >= o) incl += u;
    }
    __syncthreads();                             // wsum was last read in the digit-base scan above
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int32_t lstart = woff + incl - cnt;
    delta[tid] = dbase[tid] - lstart;
    int32_t acc = lstart;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w) {
      const int32_t c = whist[w][tid];
      whist[w][tid] = acc;
      acc += c;
    }
  }
  __syncthreads();
  // ---- local reorder: every key goes to its place in the tile's digit-sorted image ...
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r) {
    const int64_t i = wbase + r * 64 + lane;
    if (i < n) {
      const uint32_t d = (uint32_t)(k[r] >> shift) & 0xFF;
      const int32_t pos = whist[wave][d] + rank[r];
      lkey[pos] = k[r];
      lval[pos] = v[r];
    }
  }
  __syncthreads();
  // ---- ... and leaves in runs: consecutive threads hold consecutive keys of one digit => consecutive global addresses.
  // (Scattering straight from the registers made every store instruction touch up to 64 different lines: most of a
  // tile's time.)
  const int64_t tile_base = (int64_t)tile * SORT_TILE;
  const int32_t nk = (int32_t)min((int64_t)SORT_TILE, n - tile_base);
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r) {
    const int32_t p = r * SORT_BLOCK + tid;
    if (p < nk) {
      const uint64_t kk = lkey[p];
      const int64_t dst = (int64_t)delta[(uint32_t)(kk >> shift) & 0xFF] + p;
      keys_out[dst] = kk;
      vals_out[dst] = lval[p];
    }
  }
}

size_t radix_sort_scratch_bytes(int64_t n) {
  const int64_t tiles = cdiv(n > 0 ? n : 1, SORT_TILE);
  const int64_t supers = cdiv(tiles, SORT_SUPER);
  // tilehist (reused per pass) + 8 passes x (superhist + total)
  return (size_t)(tiles * 256 + 8 * (supers + 1) * 256) * sizeof(int32_t) + 1024;
}

int radix_sort_pairs(Ctx* ctx, uint64_t* keys_in, uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out,
                     int64_t n, int nbits, hipStream_t stream, const int64_t* n_dev, uint64_t** keys_res, uint32_t** vals_res) {
  EGONN_REQUIRE(n >= 0 && n < (int64_t(1) << 31), EGONN_ERR_INVALID, "radix_sort: n=%lld out of range", (long long)n);
  EGONN_REQUIRE(nbits >= 1 && nbits <= 64, EGONN_ERR_INVALID, "radix_sort: nbits=%d", nbits);
  const int passes = (nbits + 7) / 8;
  if (keys_res) *keys_res = keys_out;
  if (vals_res) *vals_res = vals_out;
  if (n == 0) return EGONN_OK;
  const int64_t tiles = cdiv(n, SORT_TILE);
  const int64_t supers = cdiv(tiles, SORT_SUPER);
  EGONN_TRY(ctx->sort_arena.ensure(radix_sort_scratch_bytes(n)));
  ctx->sort_arena.reset();
  int32_t* tilehist = ctx->sort_arena.alloc<int32_t>(tiles * 256);
  int32_t* slabs = ctx->sort_arena.alloc<int32_t>(8 * (supers + 1) * 256);
  EGONN_REQUIRE(tilehist && slabs, EGONN_ERR_STATE, "radix_sort: scratch arena too small");
  HIP_CHECK(hipMemsetAsync(slabs, 0, sizeof(int32_t) * passes * (supers + 1) * 256, stream));

  // ping-pong so that the LAST pass writes (keys_out, vals_out)
  uint64_t* kb[2] = {keys_in, keys_out};
  uint32_t* vb[2] = {vals_in, vals_out};
  int src = (passes % 2 == 1) ? 0 : 1;
  if (src == 1 && keys_res && vals_res) {   // even number of passes and a caller that takes either pair: end in the "in" pair
    src = 0;
    *keys_res = keys_in;
    *vals_res = vals_in;
  } else if (src == 1) {                    // even number of passes: start from the "out" buffers
    HIP_CHECK(hipMemcpyAsync(keys_out, keys_in, sizeof(uint64_t) * n, hipMemcpyDeviceToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(vals_out, vals_in, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, stream));
  }
  for (int p = 0; p < passes; ++p) {
    int32_t* superhist = slabs + (int64_t)p * (supers + 1) * 256;
    int32_t* total = superhist + supers * 256;
    const int shift = 8 * p;
    hipLaunchKernelGGL(sort_hist_kernel, dim3((unsigned)tiles), dim3(SORT_BLOCK), 0, stream, kb[src], n, n_dev, shift,
                       tilehist, superhist, total);
    hipLaunchKernelGGL(sort_scatter_kernel, dim3((unsigned)tiles), dim3(SORT_BLOCK), 0, stream, kb[src], vb[src],
                       kb[src ^ 1], vb[src ^ 1], n, n_dev, shift, tilehist, superhist, total);
    src ^= 1;
  }
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}


// ------------------------------------------------------------------ segmented variant (plans built from points)
// The points of a batch arrive scan by scan (offsets known on the device), so the batch index needs no sorting: every scan
// is sorted on its own Morton bits only, 9 bits per pass — 36 bits = 4 passes for coord_bits = 12 where the flat sort of
// (batch | Morton) = 40 bits needed 5 (48 -> 6 instead of 7 for coord_bits = 16).  Same two launches per pass; a tile never
// straddles two scans, the global base of digit d in a tile is
//   scan start + (keys of the scan with a smaller digit) + (keys with digit d in earlier tiles of the SAME scan),
// i.e. one row of per-scan totals and at most cdiv(n_scan, TILE) - 1 tile rows instead of the supertile pyramid.
static constexpr int SEG_BITS = 9;
static constexpr int SEG_DIGITS = 1 << SEG_BITS;          // 512: two digits per thread
static constexpr int SEG_DPT = SEG_DIGITS / SORT_BLOCK;

// tile -> (scan, first key, keys in the tile, first tile of the scan); false: the tile is beyond the last scan
__device__ static inline bool seg_locate(const int64_t* __restrict__ off, int B, int64_t n_cap, int tile, int& scan, int64_t& k0,
                                         int32_t& nk, int& tile0) {
  int t = 0;
  for (int b = 0; b < B; ++b) {
    const int64_t lo = min(off[b], n_cap), hi = min(off[b + 1], n_cap);
    const int nt = (int)((hi - lo + SORT_TILE - 1) / SORT_TILE);
    if (tile < t + nt) {
      scan = b; tile0 = t;
      k0 = lo + (int64_t)(tile - t) * SORT_TILE;
      nk = (int32_t)min((int64_t)SORT_TILE, hi - k0);
      return true;
    }
    t += nt;
  }
  return false;
}

__global__ __launch_bounds__(SORT_BLOCK) void sort_seg_hist_kernel(const uint64_t* __restrict__ keys, int64_t n_cap,
                                                                   const int64_t* __restrict__ off, int B, int shift,
                                                                   int32_t* __restrict__ tilehist, int32_t* __restrict__ scanhist) {
  __shared__ int32_t hist[SEG_DIGITS];
  const int tid = threadIdx.x, lane = tid & 63;
  int scan, tile0;
  int64_t k0;
  int32_t nk;
  if (!seg_locate(off, B, n_cap, blockIdx.x, scan, k0, nk, tile0)) return;       // workgroup-uniform
#pragma unroll
  for (int q = 0; q < SEG_DPT; ++q) hist[q * SORT_BLOCK + tid] = 0;
  __syncthreads();
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  uint64_t kreg[SORT_ROUNDS];
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r) {
    const int32_t i = r * SORT_BLOCK + tid;
    kreg[r] = (i < nk) ? keys[k0 + i] : 0ull;
  }
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r) {
    const bool valid = r * SORT_BLOCK + tid < nk;
    const uint32_t d = (uint32_t)(kreg[r] >> shift) & (SEG_DIGITS - 1);
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < SEG_BITS; ++b) {
      const bool bit = (d >> b) & 1;
      const uint64_t bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    if (valid && (m & lt) == 0) atomicAdd(&hist[d], (int32_t)__popcll(m));
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < SEG_DPT; ++q) {
    const int d = q * SORT_BLOCK + tid;
    const int32_t c = hist[d];
    tilehist[(int64_t)blockIdx.x * SEG_DIGITS + d] = c;
    if (c) atomicAdd(&scanhist[scan * SEG_DIGITS + d], c);          // <= cdiv(n_scan, TILE) adders per word
  }
}

// MODE 0: (key, value) pairs.  MODE 1 / 2: PACKED elements — one u64 = (Morton bits << idx_bits) | index of the point inside
// its scan: 8 bytes per element and pass instead of 12 (round 5).  MODE 1 moves packed elements; MODE 2 is the last pass: it
// writes the pair form the pyramid kernels read — key = (scan << key_bits) | Morton bits, value = first row of the scan + index.
template <int MODE>
__global__ __launch_bounds__(SORT_BLOCK) void sort_seg_scatter_kernel(
    const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint64_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, int64_t n_cap, const int64_t* __restrict__ off, int B, int shift,
    const int32_t* __restrict__ tilehist, const int32_t* __restrict__ scanhist, int idx_bits, int key_bits) {
  __shared__ int32_t whist[SORT_WAVES][SEG_DIGITS];   // running per-wave digit counts
  __shared__ int32_t dbase[SEG_DIGITS];               // global base of every digit for this tile
  __shared__ int32_t delta[SEG_DIGITS];               // global position - local position, per digit
  __shared__ int32_t wsum[SORT_WAVES];
  __shared__ uint64_t lkey[SORT_TILE];                // the tile, locally sorted by digit (stable)
  __shared__ uint32_t lval[SORT_TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int scan, tile0;
  int64_t k0;
  int32_t nk;
  if (!seg_locate(off, B, n_cap, blockIdx.x, scan, k0, nk, tile0)) return;       // workgroup-uniform
  const int tile = blockIdx.x;
  // the tile's keys are requested first: they do not depend on the histogram sums below, whose L2 round trips they overlap
  const int32_t wbase = wave * (64 * SORT_ROUNDS);
  uint64_t k[SORT_ROUNDS];
  uint32_t v[SORT_ROUNDS];
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r) {
    const int32_t i = wbase + r * 64 + lane;
    const bool valid = i < nk;
    k[r] = valid ? keys_in[k0 + i] : ~0ull;
    if constexpr (MODE == 0) v[r] = valid ? vals_in[k0 + i] : 0u;
    else v[r] = 0u;
  }
  // ---- global base of digit d (thread tid owns digits 2 tid, 2 tid + 1: a wave covers 128 consecutive digits)
  {
    int32_t tot[SEG_DPT], before[SEG_DPT];
#pragma unroll
    for (int q = 0; q < SEG_DPT; ++q) {
      tot[q] = scanhist[scan * SEG_DIGITS + SEG_DPT * tid + q];
      before[q] = 0;
    }
    for (int t0 = tile0; t0 < tile; t0 += 8) {           // earlier tiles of the same scan, eight rows in flight
      int32_t c[8][SEG_DPT];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int q = 0; q < SEG_DPT; ++q) c[u][q] = (t0 + u < tile) ? tilehist[(int64_t)(t0 + u) * SEG_DIGITS + SEG_DPT * tid + q] : 0;
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int q = 0; q < SEG_DPT; ++q) before[q] += c[u][q];
    }
    // exclusive scan of the scan's digit totals over the 512 digits
    int32_t mine = 0;
#pragma unroll
    for (int q = 0; q < SEG_DPT; ++q) mine += tot[q];
    int32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t u = __shfl_up(incl, o, 64);
      if (lane >= o) incl += u;
    }
    if (lane == 63) wsum[wave] = incl;
#pragma unroll
    for (int w = 0; w < SORT_WAVES; ++w)
#pragma unroll
      for (int q = 0; q < SEG_DPT; ++q) whist[w][q * SORT_BLOCK + tid] = 0;
    __syncthreads();
    int32_t run = incl - mine;
    for (int w = 0; w < wave; ++w) run += wsum[w];
    const int32_t scan_start = (int32_t)min(off[scan], n_cap);
#pragma unroll
    for (int q = 0; q < SEG_DPT; ++q) {
      dbase[SEG_DPT * tid + q] = scan_start + run + before[q];
      run += tot[q];
    }
  }
  __syncthreads();
  // ---- stable ranking: wave `wave` owns keys [wave*64*ROUNDS, +64*ROUNDS) of the tile
  int32_t rank[SORT_ROUNDS];
  volatile int32_t* my = whist[wave];
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r) {
    const bool valid = wbase + r * 64 + lane < nk;
    const uint32_t d = (uint32_t)(k[r] >> shift) & (SEG_DIGITS - 1);
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < SEG_BITS; ++b) {
      const bool bit = (d >> b) & 1;
      const uint64_t bal = __ballot(bit);
      m &= bit ? bal : ~bal;
    }
    const int32_t prior = my[d];
    __builtin_amdgcn_wave_barrier();
    rank[r] = prior + __popcll(m & lt);
    if (valid && (m & lt) == 0) my[d] = prior + __popcll(m);   // group leader publishes the new count
    __builtin_amdgcn_wave_barrier();
  }
  __syncthreads();
  // ---- per digit: the tile's count, the start of the digit's run INSIDE the tile, per wave the local position of its first key
  {
    int32_t cnt[SEG_DPT], mine = 0;
#pragma unroll
    for (int q = 0; q < SEG_DPT; ++q) {
      cnt[q] = 0;
#pragma unroll
      for (int w = 0; w < SORT_WAVES; ++w) cnt[q] += whist[w][SEG_DPT * tid + q];
      mine += cnt[q];
    }
    int32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t u = __shfl_up(incl, o, 64);
      if (lane >= o) incl += u;
    }
    __syncthreads();                             // wsum was last read in the digit-base scan above
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int32_t lstart = incl - mine;
    for (int w = 0; w < wave; ++w) lstart += wsum[w];
#pragma unroll
    for (int q = 0; q < SEG_DPT; ++q) {
      const int d = SEG_DPT * tid + q;
      delta[d] = dbase[d] - lstart;
      int32_t acc = lstart;
#pragma unroll
      for (int w = 0; w < SORT_WAVES; ++w) {
        const int32_t c = whist[w][d];
        whist[w][d] = acc;
        acc += c;
      }
      lstart += cnt[q];
    }
  }
  __syncthreads();
  // ---- local reorder, then leave in runs (consecutive threads = consecutive keys of one digit = consecutive addresses)
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r) {
    if (wbase + r * 64 + lane < nk) {
      const uint32_t d = (uint32_t)(k[r] >> shift) & (SEG_DIGITS - 1);
      const int32_t pos = whist[wave][d] + rank[r];
      lkey[pos] = k[r];
      if constexpr (MODE == 0) lval[pos] = v[r];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < SORT_ROUNDS; ++r) {
    const int32_t p = r * SORT_BLOCK + tid;
    if (p < nk) {
      const uint64_t kk = lkey[p];
      const int64_t dst = (int64_t)delta[(uint32_t)(kk >> shift) & (SEG_DIGITS - 1)] + p;
      if constexpr (MODE == 0) {
        keys_out[dst] = kk;
        vals_out[dst] = lval[p];
      } else if constexpr (MODE == 1) {
        keys_out[dst] = kk;
      } else {
        keys_out[dst] = ((uint64_t)scan << key_bits) | (kk >> idx_bits);
        vals_out[dst] = (uint32_t)(kk & ((1ull << idx_bits) - 1)) + (uint32_t)min(off[scan], n_cap);
      }
    }
  }
}

// scratch of the segmented sort inside ctx->sort_arena (deterministic for (n, B)): per-tile and per-scan digit histograms
int radix_sort_segments_layout(Ctx* ctx, int64_t n, int B, int32_t** tilehist, int32_t** scanhist) {
  const int64_t tiles = cdiv(n > 0 ? n : 1, SORT_TILE) + B;
  EGONN_TRY(ctx->sort_arena.ensure(radix_sort_segments_scratch_bytes(n, B)));
  ctx->sort_arena.reset();
  *tilehist = ctx->sort_arena.alloc<int32_t>(tiles * SEG_DIGITS);
  *scanhist = ctx->sort_arena.alloc<int32_t>((size_t)8 * B * SEG_DIGITS);
  EGONN_REQUIRE(*tilehist && *scanhist, EGONN_ERR_STATE, "radix_sort: scratch arena too small");
  return EGONN_OK;
}
int radix_sort_segments_passes(int nbits) { return (nbits + SEG_BITS - 1) / SEG_BITS; }
int radix_sort_segments_digits() { return SEG_DIGITS; }

size_t radix_sort_segments_scratch_bytes(int64_t n, int B) {
  const int64_t tiles = cdiv(n > 0 ? n : 1, SORT_TILE) + B;
  return (size_t)(tiles * SEG_DIGITS + 8 * (int64_t)B * SEG_DIGITS) * sizeof(int32_t) + 1024;
}

// Sorts every scan [off[b], off[b+1]) of (keys, vals) on bits [0, nbits) of the key (stable).  off: DEVICE int64 (B+1), clipped
// to n (the capacity the buffers and the grid are sized for).  Result: as radix_sort_pairs (keys_res / vals_res).
// idx_bits > 0: keys_in holds PACKED elements (Morton bits << idx_bits | index inside the scan; vals_in is not read): every pass
// but the last moves 8 bytes per element, the last one writes the (key | scan, value) pairs described above.
int radix_sort_segments(Ctx* ctx, uint64_t* keys_in, uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out, int64_t n,
                        const int64_t* off_dev, int B, int nbits, hipStream_t stream, uint64_t** keys_res, uint32_t** vals_res,
                        int idx_bits) {
  EGONN_REQUIRE(n >= 0 && n < (int64_t(1) << 31), EGONN_ERR_INVALID, "radix_sort: n=%lld out of range", (long long)n);
  EGONN_REQUIRE(nbits >= 1 && nbits <= 64 && off_dev && B >= 1 && idx_bits >= 0 && nbits + idx_bits <= 64, EGONN_ERR_INVALID,
                "radix_sort_segments: bad arguments");
  const int passes = (nbits + SEG_BITS - 1) / SEG_BITS;
  EGONN_REQUIRE(passes <= 8, EGONN_ERR_INVALID, "radix_sort_segments: %d passes", passes);
  *keys_res = keys_out;
  *vals_res = vals_out;
  if (n == 0) return EGONN_OK;
  const int64_t tiles = cdiv(n, SORT_TILE) + B;
  int32_t *tilehist = nullptr, *scanhist = nullptr;
  EGONN_TRY(radix_sort_segments_layout(ctx, n, B, &tilehist, &scanhist));
  // (plans built from points: the key kernel in front of the sort has zeroed the per-scan histograms already — one memset node less)
  if (ctx->sort_prezeroed != scanhist) HIP_CHECK(hipMemsetAsync(scanhist, 0, sizeof(int32_t) * passes * B * SEG_DIGITS, stream));
  ctx->sort_prezeroed = nullptr;
  uint64_t* kb[2] = {keys_in, keys_out};
  uint32_t* vb[2] = {vals_in, vals_out};
  int src = 0;
  for (int p = 0; p < passes; ++p) {
    int32_t* sh = scanhist + (int64_t)p * B * SEG_DIGITS;
    const int shift = SEG_BITS * p + idx_bits;
    hipLaunchKernelGGL(sort_seg_hist_kernel, dim3((unsigned)tiles), dim3(SORT_BLOCK), 0, stream, kb[src], n, off_dev, B, shift, tilehist, sh);
    if (idx_bits == 0)
      hipLaunchKernelGGL(sort_seg_scatter_kernel<0>, dim3((unsigned)tiles), dim3(SORT_BLOCK), 0, stream, kb[src], vb[src], kb[src ^ 1],
                         vb[src ^ 1], n, off_dev, B, shift, tilehist, sh, 0, nbits);
    else if (p + 1 < passes)
      hipLaunchKernelGGL(sort_seg_scatter_kernel<1>, dim3((unsigned)tiles), dim3(SORT_BLOCK), 0, stream, kb[src], vb[src], kb[src ^ 1],
                         vb[src ^ 1], n, off_dev, B, shift, tilehist, sh, idx_bits, nbits);
    else
      hipLaunchKernelGGL(sort_seg_scatter_kernel<2>, dim3((unsigned)tiles), dim3(SORT_BLOCK), 0, stream, kb[src], vb[src], kb[src ^ 1],
                         vb[src ^ 1], n, off_dev, B, shift, tilehist, sh, idx_bits, nbits);
    src ^= 1;
  }
  *keys_res = kb[src];                                    // whichever pair the last pass wrote
  *vals_res = vb[src];
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

}  // namespace egonn
