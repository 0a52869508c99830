// Coordinate plan: voxelisation, Z-order keys, unique, the stride-2 pyramid, 4x4x4 occupancy
// masks and the kernel maps (k=3 neighbour tables, k=2/s=2 child tables, transposed tables).
//
// Replaces (reference call sites): ME.utils.sparse_quantize (datasets/quantization.py:42,83),
// ME.utils.batched_coordinates (eval/evaluate.py:333), ME.SparseTensor's coordinate-map build
// (models/minkgl.py:269) and ME's coordinate manager (strided maps + kernel maps) used by every
// MinkowskiConvolution / MinkowskiConvolutionTranspose in models/minkgl.py.
//
// One host synchronisation per plan (the size query): after sort + pyramid, the per-level row counts
// and per-sample offsets are copied to pinned host memory; everything after that is launched with
// exact grids.
#include "common.h"

#include <algorithm>
#include <stdarg.h>

namespace egonn {

// ------------------------------------------------------------------ error + arena
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_err; }

int Arena::ensure(size_t bytes) {
  if (bytes <= cap) return EGONN_OK;
  size_t want = align_up(bytes + bytes / 4, size_t(1) << 20);
  if (base) {
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipFree(base));
    base = nullptr;
    cap = 0;
  }
  HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&base), want));
  cap = want;
  off = 0;
  return EGONN_OK;
}
void Arena::release() {
  if (base) (void)hipFree(base);
  base = nullptr;
  cap = off = 0;
}

hipEvent_t Profiler::get() {
  if (!pool.empty()) {
    hipEvent_t e = pool.back();
    pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
ProfScope::ProfScope(Ctx* c, hipStream_t s, const char* name, int kind, int level, int K, int cin, int cout, int es)
    : ctx(c), st(s) {
  if (!c || c->prof.mode == 0) return;
  if (c->prof.mode >= 2 && !strstr(name, c->prof.filter)) return;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cs);
  in_graph = cs == hipStreamCaptureStatusActive;
  if (in_graph && c->prof.mode != 3) return;            // only the event-record brackets can live inside a graph
  ProfRec r;
  snprintf(r.name, sizeof(r.name), "%s", name);
  r.kind = kind; r.level = level; r.K = K; r.cin = cin; r.cout = cout; r.es = es;
  r.e0 = c->prof.get();
  r.e1 = c->prof.get();
  exact = c->prof.mode == 2 && (kind == PK_K3 || kind == PK_K2S2 || kind == PK_TCONV);
  if (exact) {
    prof_kernel_events()[0] = r.e0;
    prof_kernel_events()[1] = r.e1;
  } else {
    (void)hipEventRecord(r.e0, s);
  }
  std::vector<ProfRec>& dst = in_graph ? c->prof.graph_recs : c->prof.recs;
  idx = (int)dst.size();
  dst.push_back(r);
}
ProfScope::~ProfScope() {
  if (idx < 0) return;
  if (exact) {
    prof_kernel_events()[0] = prof_kernel_events()[1] = nullptr;
  } else {
    (void)hipEventRecord((in_graph ? ctx->prof.graph_recs : ctx->prof.recs)[idx].e1, st);
  }
}
hipEvent_t* prof_kernel_events() {
  static thread_local hipEvent_t ev[2] = {nullptr, nullptr};
  return ev;
}

// ------------------------------------------------------------------ key construction
struct QuantParams {
  int mode;        // 0 = cartesian floor(p / q), 1 = polar (theta deg, r, z) / (s0, s1, s2)
  float s0, s1, s2;
};

// floor(p / q) in IEEE fp32 — `torch.floor(pc / q)` on CPU is a true fp32 division (pinned in
// tests/test_oracle.py::test_cartesian_division_is_true_fp32_division).  The __f*_rn intrinsics stop
// the compiler from contracting or reassociating.
__device__ static inline void quantize_point(const QuantParams& qp, float x, float y, float z, int32_t& cx,
                                             int32_t& cy, int32_t& cz) {
  if (qp.mode == 0) {
    cx = (int32_t)floorf(qp.s0 == 1.0f ? x : __fdiv_rn(x, qp.s0));
    cy = (int32_t)floorf(qp.s0 == 1.0f ? y : __fdiv_rn(y, qp.s0));
    cz = (int32_t)floorf(qp.s0 == 1.0f ? z : __fdiv_rn(z, qp.s0));
  } else {
    // reference datasets/quantization.py:35-40, evaluated left to right in fp32.  atan2 is taken in fp64 and rounded ONCE
    // to fp32: a correctly rounded fp32 arc tangent, which is what the CPU libraries (glibc / numpy; torch's sleef to
    // 1 ulp) return — ocml's fp32 atan2f differs from them by an ulp on ~0.1 % of the points, enough to move a point
    // across a sector boundary.
    const float at = (float)atan2((double)y, (double)x);
    const float theta = __fadd_rn(180.0f, __fdiv_rn(__fmul_rn(at, 180.0f), 3.14159265358979323846f));
    const float dist = sqrtf(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)));
    cx = (int32_t)floorf(__fdiv_rn(theta, qp.s0));
    cy = (int32_t)floorf(__fdiv_rn(dist, qp.s1));
    cz = (int32_t)floorf(__fdiv_rn(z, qp.s2));
  }
}

__device__ static inline bool encode_key(int32_t b, int32_t cx, int32_t cy, int32_t cz, int cb, uint64_t& key) {
  const int32_t bias = 1 << (cb - 1);
  const int32_t lim = 1 << cb;
  const int32_t ux = cx + bias, uy = cy + bias, uz = cz + bias;
  const bool ok = (ux >= 0) & (ux < lim) & (uy >= 0) & (uy < lim) & (uz >= 0) & (uz < lim) & (b >= 0);
  key = ((uint64_t)(uint32_t)b << (3 * cb)) | morton3((uint32_t)ux, (uint32_t)uy, (uint32_t)uz);
  return ok;
}

// n_cap sizes the grid; the true point count is scan_off[B] (device memory: capturable plans never tell the host)
// idx_bits > 0: PACKED output for the segmented sort (sort.hip) — keys[i] = (Morton bits << idx_bits) | (i - first point of
// the scan), no value array: the batch index is implied by the segment and restored by the sort's last pass.
__global__ void points_to_keys_kernel(const float* __restrict__ pts, int64_t n_cap, const int64_t* __restrict__ scan_off,
                                      int B, QuantParams qp, int cb, uint64_t* __restrict__ keys,
                                      uint32_t* __restrict__ vals, int32_t* __restrict__ flags, int idx_bits,
                                      int64_t* __restrict__ off_copy, int32_t* __restrict__ zero_ptr, int zero_words) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // riders (graph nodes saved): the plan's own copy of the scan offsets, and the zeroing of the sort's per-scan histograms —
  // both are read by LATER launches only
  if (off_copy && i <= B) off_copy[i] = scan_off[i];
  for (int64_t w = i; w < zero_words; w += (int64_t)gridDim.x * blockDim.x) zero_ptr[w] = 0;
  const int64_t nn = scan_off[B];
  if (i == 0 && nn > n_cap) atomicOr(flags, 2);         // more points than the plan was reserved for
  // the segmented sort moves only rows inside [off[b], off[b+1]): offsets that do not start at 0 or that decrease would leave
  // rows unsorted (stale buffer contents) -> reported as a range error (egonn_voxelize raises, egonn_plan_status for reserved plans)
  if (i <= B && ((i == 0 && scan_off[0] != 0) || (i > 0 && scan_off[i] < scan_off[i - 1]))) atomicOr(flags, 1);
  if (i >= (nn < n_cap ? nn : n_cap)) return;
  // sample index = last b with scan_off[b] <= i
  int lo = 0, hi = B;   // invariant: scan_off[lo] <= i < scan_off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (scan_off[mid] <= i) lo = mid; else hi = mid;
  }
  const float x = pts[3 * i + 0], y = pts[3 * i + 1], z = pts[3 * i + 2];
  int32_t cx, cy, cz;
  quantize_point(qp, x, y, z, cx, cy, cz);
  uint64_t key;
  const bool finite = isfinite(x) & isfinite(y) & isfinite(z);
  if (!encode_key(lo, cx, cy, cz, cb, key) || !finite) {
    // out of the coordinate range / not a number: reported (egonn_voxelize raises EGONN_ERR_RANGE, egonn_plan_status for
    // reserved plans), and the point is CLAMPED into its own sample's range so that every later kernel of a reserved
    // (sync-free) plan still sees a well-formed key: a sentinel key would carry a sample index far beyond the batch and
    // index the per-sample arrays out of bounds (found by tests/test_gpu_parity.py::test_streaming_pipeline_equals_extract)
    atomicOr(flags, 1);
    const int32_t hi = (1 << (cb - 1)) - 1, lo_c = -(1 << (cb - 1));
    const int32_t qx = finite ? min(max(cx, lo_c), hi) : 0, qy = finite ? min(max(cy, lo_c), hi) : 0,
                  qz = finite ? min(max(cz, lo_c), hi) : 0;
    encode_key(lo, qx, qy, qz, cb, key);
  }
  if (idx_bits > 0) {
    keys[i] = ((key & ((1ull << (3 * cb)) - 1)) << idx_bits) | (uint64_t)(i - scan_off[lo]);
  } else {
    keys[i] = key;
    vals[i] = (uint32_t)i;
  }
}

__global__ void coords_to_keys_kernel(const int32_t* __restrict__ c4, int64_t n, int cb, int Bmax,
                                      uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                      int32_t* __restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = reinterpret_cast<const int4*>(c4)[i];
  uint64_t key;
  if (!encode_key(c.x, c.y, c.z, c.w, cb, key) || c.x >= Bmax) {
    atomicOr(flags, 1);
    key = ~0ull >> 1;
  }
  keys[i] = key;
  vals[i] = (uint32_t)i;
}

// ------------------------------------------------------------------ pyramid (all levels in two launches)
// For sorted keys, element i starts a new group at level l  iff  (key_i >> 3l) != (key_{i-1} >> 3l).
// h_i = number of levels (0..NL-1) at which i is a group head; heads are nested (head at l => head at l-1).
static constexpr int PYR_BLOCK = 256;
static constexpr int PYR_ROUNDS = 4;     // keys per lane: 8 / 4 / 2 -> count + apply 32.6 / 23.9 / 23.2 us at batch 16 (round 6)
static constexpr int PYR_TILE = PYR_BLOCK * PYR_ROUNDS;
static constexpr int NL = EGONN_MAX_LEVELS;

__device__ static inline int head_levels(const uint64_t* __restrict__ keys, int64_t i) {
  if (i == 0) return NL;
  const uint64_t d = keys[i] ^ keys[i - 1];
  if (d == 0) return 0;
  const int t = (63 - __clzll(d)) / 3;
  return (t >= NL - 1) ? NL : t + 1;
}

__device__ static inline int64_t clip_count(int64_t cap, const int64_t* __restrict__ n_dev) {
  if (!n_dev) return cap;
  const int64_t v = *n_dev;
  return v < cap ? v : cap;
}
__global__ __launch_bounds__(PYR_BLOCK) void pyramid_count_kernel(const uint64_t* __restrict__ keys, int64_t n_cap,
                                                                  const int64_t* __restrict__ n_dev,
                                                                  int32_t* __restrict__ tilecnt) {
  __shared__ int32_t cnt[PYR_BLOCK / 64][NL];
  const int64_t n = clip_count(n_cap, n_dev);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t wbase = (int64_t)blockIdx.x * PYR_TILE + (int64_t)wave * 64 * PYR_ROUNDS;
  int32_t c[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) c[l] = 0;
  for (int r = 0; r < PYR_ROUNDS; ++r) {
    const int64_t i = wbase + r * 64 + lane;
    const int h = (i < n) ? head_levels(keys, i) : 0;
#pragma unroll
    for (int l = 0; l < NL; ++l) c[l] += __popcll(__ballot(h > l));
  }
  if (lane == 0) {
#pragma unroll
    for (int l = 0; l < NL; ++l) cnt[wave][l] = c[l];
  }
  __syncthreads();
  if (tid < NL) {
    int32_t s = 0;
    for (int w = 0; w < PYR_BLOCK / 64; ++w) s += cnt[w][tid];
    tilecnt[(int64_t)blockIdx.x * NL + tid] = s;
  }
}

struct PyramidOut {
  uint64_t* keys[NL];
  int32_t* parent[NL];
  int32_t* cstart[NL];
  int32_t* boff[EGONN_NUM_LEVELS];
  int32_t* perm0;
  int32_t* counts;    // [NL] rows per level, [NL] = batch size seen (max batch index + 1), [NL+1] = input rows
  int32_t* flags;     // bit 1: a level has more rows than its capacity
  int32_t cap[NL];    // row capacity of every level (maps and workspaces are sized for it)
};

__global__ __launch_bounds__(PYR_BLOCK) void pyramid_apply_kernel(const uint64_t* __restrict__ keys,
                                                                  const uint32_t* __restrict__ vals, int64_t n_cap,
                                                                  const int64_t* __restrict__ n_dev,
                                                                  const int32_t* __restrict__ tilecnt, int ntiles,
                                                                  int cb, int B, PyramidOut out) {
  __shared__ int32_t red[PYR_BLOCK / 64][NL];
  const int64_t n = clip_count(n_cap, n_dev);
  __shared__ int32_t tilebase[NL];
  __shared__ int32_t wavecnt[PYR_BLOCK / 64][NL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tile = blockIdx.x;

  // ---- rows of every level in earlier tiles
  {
    int32_t p[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) p[l] = 0;
    for (int t = tid; t < tile; t += PYR_BLOCK) {
#pragma unroll
      for (int l = 0; l < NL; ++l) p[l] += tilecnt[(int64_t)t * NL + l];
    }
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      int32_t v = p[l];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
      if (lane == 0) red[wave][l] = v;
    }
    __syncthreads();
    if (tid < NL) {
      int32_t s = 0;
      for (int w = 0; w < PYR_BLOCK / 64; ++w) s += red[w][tid];
      tilebase[tid] = s;
    }
  }

  // ---- pass 1: head levels of this wave's keys + wave totals
  const int64_t wbase = (int64_t)tile * PYR_TILE + (int64_t)wave * 64 * PYR_ROUNDS;
  int hreg[PYR_ROUNDS];
  int32_t wc[NL];
#pragma unroll
  for (int l = 0; l < NL; ++l) wc[l] = 0;
#pragma unroll
  for (int r = 0; r < PYR_ROUNDS; ++r) {
    const int64_t i = wbase + r * 64 + lane;
    hreg[r] = (i < n) ? head_levels(keys, i) : 0;
#pragma unroll
    for (int l = 0; l < NL; ++l) wc[l] += __popcll(__ballot(hreg[r] > l));
  }
  if (lane == 0) {
#pragma unroll
    for (int l = 0; l < NL; ++l) wavecnt[wave][l] = wc[l];
  }
  __syncthreads();
  int32_t run[NL];   // heads of level l strictly before the current round of this wave (wave-uniform)
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    int32_t s = tilebase[l];
    for (int w = 0; w < wave; ++w) s += wavecnt[w][l];
    run[l] = s;
  }

  // ---- pass 2: rank and write
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int bshift = 3 * cb;
#pragma unroll
  for (int r = 0; r < PYR_ROUNDS; ++r) {
    const int64_t i = wbase + r * 64 + lane;
    const int h = hreg[r];
    const uint64_t key = (i < n) ? keys[i] : 0ull;
    int32_t excl[NL];   // heads of level l among elements < i
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const uint64_t m = __ballot(h > l);
      excl[l] = run[l] + __popcll(m & lt);
      run[l] += __popcll(m);
    }
    // A batch that exceeds the reservation of a level: rows beyond the capacity are NOT written, and every index that is
    // written (first child, parent, per-sample offset) is clipped to the capacity of the level it points into, so that all
    // later kernels of a reserved (sync-free) plan stay inside their tables; the overflow is flagged (egonn_plan_status).
    if (i < n && h > 0) {
      if (excl[0] < out.cap[0]) out.perm0[excl[0]] = (int32_t)vals[i];
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        if (l < h && excl[l] < out.cap[l]) {
          out.keys[l][excl[l]] = key >> (3 * l);
          if (l >= 1) out.cstart[l][excl[l]] = min(excl[l - 1], out.cap[l - 1]);
          if (l + 1 < NL) {
            // group index of i at level l+1: a head there -> excl, otherwise the group opened earlier
            const int32_t g = (l + 1 < h) ? excl[l + 1] : excl[l + 1] - 1;
            out.parent[l][excl[l]] = min(g, out.cap[l + 1] - 1);
          }
        }
      }
      // per-sample offsets: a change of the batch index makes i a head at every level
      const int32_t b = (int32_t)(key >> bshift);
      const int32_t bprev = (i == 0) ? -1 : (int32_t)(keys[i - 1] >> bshift);
      if (b != bprev) {
        for (int32_t bb = bprev + 1; bb <= b && bb <= B; ++bb) {
#pragma unroll
          for (int l = 0; l < EGONN_NUM_LEVELS; ++l) out.boff[l][bb] = min(excl[l], out.cap[l]);
        }
      }
    }
    // ---- the globally last element closes every table
    if (i == n - 1) {
      const int32_t blast = (int32_t)(key >> bshift);
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        const int32_t total = excl[l] + (h > l ? 1 : 0);
        out.counts[l] = total;
        if (total > out.cap[l]) atomicOr(out.flags, 2);
        if (l >= 1) {
          const int32_t below = excl[l - 1] + (h > l - 1 ? 1 : 0);
          out.cstart[l][min(total, out.cap[l])] = min(below, out.cap[l - 1]);
        }
        if (l < EGONN_NUM_LEVELS) {
          for (int32_t bb = blast + 1; bb <= B; ++bb) out.boff[l][bb] = min(total, out.cap[l]);
        }
      }
      out.counts[NL] = blast + 1;
      out.counts[NL + 1] = (int32_t)n;
    }
  }
}

// ------------------------------------------------------------------ 4x4x4 occupancy masks
// Level L row j is a 4x4x4 block of level-(L-2) voxels; its rows at level L-2 are contiguous.
struct MaskArgs {
  const int32_t* cstartL[NL];     // cstart of level L
  const uint64_t* keys[NL];       // keys of level L
  uint64_t* mask[NL];
  int32_t* bstart[NL];
  const int32_t* counts;          // device row counts per level
  int32_t prefix[NL + 1];         // prefix over levels 2..NL-1 of the level capacities (grid layout)
};

__device__ static inline int32_t level_rows(const int32_t* __restrict__ counts, int l, int32_t cap) {
  const int32_t v = counts[l];
  return v < cap ? v : cap;
}

__device__ static inline void block_mask_body(const MaskArgs& a, uint32_t vblock) {
  const int64_t t = (int64_t)vblock * 256 + threadIdx.x;
  if (t >= a.prefix[NL]) return;
  int L = 2;
  while (L < NL - 1 && t >= a.prefix[L + 1]) ++L;
  const int32_t j = (int32_t)(t - a.prefix[L]);
  if (j >= level_rows(a.counts, L, a.prefix[L + 1] - a.prefix[L])) return;
  const int32_t* cs1 = a.cstartL[L];
  const int32_t* cs2 = a.cstartL[L - 1];
  const int32_t s = cs2[cs1[j]];
  const int32_t e = cs2[cs1[j + 1]];
  const uint64_t* k = a.keys[L - 2];
  uint64_t m = 0;
  for (int32_t r = s; r < e; ++r) m |= 1ull << (k[r] & 63);
  a.mask[L][j] = m;
  a.bstart[L][j] = s;
}

// ------------------------------------------------------------------ k=3 neighbour table
// One wave per level-(l+2) block.  27 lanes find the adjacent blocks by binary search; every
// (voxel, offset) pair is then an LDS mask test + popcount.
__device__ static inline int32_t find_key(const uint64_t* __restrict__ keys, int32_t n, uint64_t q) {
  int32_t lo = 0, hi = n;
  while (lo < hi) {
    const int32_t mid = (lo + hi) >> 1;
    if (keys[mid] < q) lo = mid + 1; else hi = mid;
  }
  return (lo < n && keys[lo] == q) ? lo : -1;
}

__device__ static inline uint32_t bit_of_local(uint32_t lx, uint32_t ly, uint32_t lz) {   // 2 bits each
  return (lx & 1) | ((ly & 1) << 1) | ((lz & 1) << 2) | ((lx & 2) << 2) | ((ly & 2) << 3) | ((lz & 2) << 4);
}

__device__ static inline int32_t lookup_local(const uint64_t* nbm, const int32_t* nbs, int32_t nx, int32_t ny,
                                              int32_t nz) {
  // nx,ny,nz in [-2, 5]: local coordinates relative to the centre block
  const int32_t sx = (nx < 0) ? 0 : (nx > 3 ? 2 : 1);
  const int32_t sy = (ny < 0) ? 0 : (ny > 3 ? 2 : 1);
  const int32_t sz = (nz < 0) ? 0 : (nz > 3 ? 2 : 1);
  const int32_t slot = sx + 3 * sy + 9 * sz;
  const uint32_t bit = bit_of_local((uint32_t)nx & 3, (uint32_t)ny & 3, (uint32_t)nz & 3);
  const uint64_t m = nbm[slot];
  if (!((m >> bit) & 1)) return -1;
  return nbs[slot] + __popcll(m & ((1ull << bit) - 1));
}

// Row adjacency by direct search: adj[i][s] = row of the same-level voxel at offset s (27 slots, x fastest), -1 =
// absent.  Used for the two top (virtual) levels only, where N is tiny; every level below derives its table
// from the table two levels up (nbr27_kernel), so no level with many rows ever binary-searches.
// (the two searched levels are independent: one launch covers both; threads past the first level's cap0 * 27 slots
//  belong to the second)
struct Adj27Args {
  const uint64_t *keys0, *keys1;
  const int32_t* counts;
  int level0, level1, cbL0, cbL1;
  int32_t cap0, cap1;
  int32_t *adj0, *adj1;
};
__device__ static inline void adj27_search_body(const Adj27Args& a, uint32_t vblock) {
  const uint64_t* __restrict__ keys0 = a.keys0;
  const uint64_t* __restrict__ keys1 = a.keys1;
  const int32_t* __restrict__ counts = a.counts;
  const int level0 = a.level0, level1 = a.level1, cbL0 = a.cbL0, cbL1 = a.cbL1;
  const int32_t cap0 = a.cap0, cap1 = a.cap1;
  int32_t* __restrict__ adj0 = a.adj0;
  int32_t* __restrict__ adj1 = a.adj1;
  int32_t t = (int32_t)(vblock * 256 + threadIdx.x);
  const bool second = t >= cap0 * 27;
  if (second) t -= cap0 * 27;
  const uint64_t* __restrict__ keys = second ? keys1 : keys0;
  int32_t* __restrict__ adj = second ? adj1 : adj0;
  const int level = second ? level1 : level0, cbL = second ? cbL1 : cbL0;
  const int32_t cap = second ? cap1 : cap0;
  const int32_t n = level_rows(counts, level, cap);
  if (t >= n * 27) return;
  const int32_t i = t / 27, sl = t - i * 27;
  const uint64_t key = keys[i];
  const uint64_t mort = key & ((1ull << (3 * cbL)) - 1);
  const uint64_t bat = key >> (3 * cbL);
  const int32_t nx = (int32_t)compact1by2(mort) + (sl % 3) - 1, ny = (int32_t)compact1by2(mort >> 1) + (sl / 3) % 3 - 1,
                nz = (int32_t)compact1by2(mort >> 2) + sl / 9 - 1;
  const int32_t lim = 1 << cbL;
  int32_t r = -1;
  if (sl == 13) r = i;
  else if (nx >= 0 && nx < lim && ny >= 0 && ny < lim && nz >= 0 && nz < lim)
    r = find_key(keys, n, (bat << (3 * cbL)) | morton3((uint32_t)nx, (uint32_t)ny, (uint32_t)nz));
  adj[t] = r;
}

// k=3 neighbour table of level l from the adjacency of its 4x4x4 blocks (= the k=3 table of level l+2).
// One wave per block: 27 lanes fetch (mask, first row) of the adjacent blocks into LDS, then every
// (voxel, offset) pair is an LDS mask test + popcount.
// Two levels per launch: level l needs the table of level l+2 only, so (l, l-1) are independent and share a launch
// (seven ~5 us launches per step became four).
struct Nbr27Job {
  const uint64_t* vkeys;       // level l
  const int32_t* badj;         // [nblocks][27] level l+2
  const uint64_t* bmask;
  const int32_t* bstart;
  int32_t* nbr;
  int32_t level, cap_blocks, cap_vox;
};
struct Blk27Args {              // first-layer (k=5) helper: per level-2 block the (mask, first row) of its 27 neighbours
  const int32_t* badj;
  const uint64_t* bmask;
  const int32_t* bstart;
  int32_t cap2;
  uint64_t* t2m;
  int32_t* t2s;
};
struct Nbr27Pair {
  Nbr27Job job[2];
  const int32_t* counts;
  uint32_t split;              // workgroups of job[0]
  uint32_t tail_at;            // first workgroup of the blk27 rider (the launch of level 1 carries it: it reads the level-2
  Blk27Args blk;               //  table finished by the launch before); ~0u = none
};
__device__ static inline void blk27_body(const Blk27Args& b, const int32_t* __restrict__ counts, uint32_t vblock) {
  const int32_t t = (int32_t)(vblock * 256 + threadIdx.x);
  if (t >= level_rows(counts, 2, b.cap2) * 27) return;
  const int32_t a = b.badj[t];
  b.t2m[t] = a >= 0 ? b.bmask[a] : 0ull;
  b.t2s[t] = a >= 0 ? b.bstart[a] : 0;
}
__global__ __launch_bounds__(256) void nbr27_kernel(const Nbr27Pair a) {
  if (blockIdx.x >= a.tail_at) {                       // workgroup-uniform
    blk27_body(a.blk, a.counts, blockIdx.x - a.tail_at);
    return;
  }
  const bool second = blockIdx.x >= a.split;
  const Nbr27Job& J = a.job[second ? 1 : 0];
  const uint64_t* __restrict__ vkeys = J.vkeys;
  const int32_t* __restrict__ badj = J.badj;
  const uint64_t* __restrict__ bmask = J.bmask;
  const int32_t* __restrict__ bstart = J.bstart;
  const int32_t* __restrict__ counts = a.counts;
  int32_t* __restrict__ nbr = J.nbr;
  const int level = J.level;
  const int32_t cap_blocks = J.cap_blocks, cap_vox = J.cap_vox;
  const uint32_t bid = second ? blockIdx.x - a.split : blockIdx.x;
  __shared__ uint64_t s_m[4][27];
  __shared__ int32_t s_s[4][27];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int32_t j = (int32_t)bid * 4 + wave;
  const int32_t nblocks = level_rows(counts, level + 2, cap_blocks), nvox = level_rows(counts, level, cap_vox);
  if (j >= nblocks) return;
  // Two memory round trips per block instead of four: (adjacency row, first/last voxel row of the block) are requested
  // together, then (masks + first rows of the 27 adjacent blocks, the block's voxel keys) together.  The kernel is a chain
  // of latencies at full occupancy (a block has ~6 voxels), so the number of dependent trips is its run time.
  // (Measured and rejected: persistent workgroups with the loads issued one block ahead and a (position, offset) lookup
  // table in LDS — 70 vs 62 us per step at batch 16, 136 vs 143 at batch 64.)
  // (Round 5, also rejected: two blocks per wave — lanes 0-26 / 32-58 fetch two adjacency rows together: 52 -> 59 us per step;
  // with 32 waves per CU the two round trips are hidden already, what counts is the item loop, which then runs twice per wave.)
  // (clamped: when level `level` overflows its reservation but level + 2 does not, bstart still holds unclipped rows
  //  and the table write below would leave the cap_vox x 27 allocation)
  const int32_t s = min(bstart[j], nvox);
  const int32_t e = min((j + 1 < nblocks) ? bstart[j + 1] : nvox, nvox);
  const int32_t idx = (lane < 27) ? badj[(int64_t)j * 27 + lane] : -1;
  const int32_t items = (e - s) * 27;
  uint64_t key0 = 0;                                      // key of the voxel of this lane's first (voxel, offset) slot
  if (lane < items) key0 = vkeys[s + lane / 27];
  if (lane < 27) {
    s_m[wave][lane] = idx >= 0 ? bmask[idx] : 0ull;
    s_s[wave][lane] = idx >= 0 ? bstart[idx] : 0;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  for (int32_t t = lane; t < items; t += 64) {
    const int32_t v = t / 27, k = t - v * 27;
    const uint32_t lk = (uint32_t)((t == lane ? key0 : vkeys[s + v]) & 63);
    const int32_t lx = (lk & 1) | ((lk >> 2) & 2), ly = ((lk >> 1) & 1) | ((lk >> 3) & 2),
                  lz = ((lk >> 2) & 1) | ((lk >> 4) & 2);
    const int32_t nx = lx + (k % 3) - 1, ny = ly + (k / 3) % 3 - 1, nz = lz + k / 9 - 1;
    nbr[(int64_t)s * 27 + t] = lookup_local(s_m[wave], s_s[wave], nx, ny, nz);
  }
}

// Number of valid entries of a kernel map (pairs P of SURVEY.md §8d).  Only launched by egonn_profile_fetch —
// a same-address atomic per wave inside the map builder costs ~12 ns each and dominated it (profiles/r01b).
__global__ void count_valid_kernel(const int32_t* __restrict__ tbl, int64_t n, unsigned long long* __restrict__ out) {
  __shared__ int32_t red[4];
  int32_t c = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    c += (tbl[i] >= 0);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (unsigned long long)(red[0] + red[1] + red[2] + red[3]));
}
// pairs of the never-materialised 5x5x5 map of the first layer: for every level-0 voxel, the occupied voxels among its
// 125 neighbours = bits of the 27 surrounding 4x4x4 block masks inside the voxel's 5x5x5 box (profiling only)
__global__ void count_k5_pairs_kernel(const uint64_t* __restrict__ vkeys, const int32_t* __restrict__ g0,
                                      const uint64_t* __restrict__ t2m, int32_t n, unsigned long long* __restrict__ out) {
  __shared__ int32_t red[4];
  int32_t c = 0;
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t lk = (uint32_t)(vkeys[i] & 63);
    const int lx = (lk & 1) | ((lk >> 2) & 2), ly = ((lk >> 1) & 1) | ((lk >> 3) & 2), lz = ((lk >> 2) & 1) | ((lk >> 4) & 2);
    const uint64_t* m27 = t2m + (int64_t)g0[i] * 27;
    for (int k = 0; k < 125; ++k) {
      const int nx = lx + k % 5 - 2, ny = ly + (k / 5) % 5 - 2, nz = lz + k / 25 - 2;
      const int slot = ((nx < 0) ? 0 : (nx > 3 ? 2 : 1)) + 3 * ((ny < 0) ? 0 : (ny > 3 ? 2 : 1)) + 9 * ((nz < 0) ? 0 : (nz > 3 ? 2 : 1));
      c += (int32_t)((m27[slot] >> bit_of_local((uint32_t)nx & 3, (uint32_t)ny & 3, (uint32_t)nz & 3)) & 1);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (unsigned long long)(red[0] + red[1] + red[2] + red[3]));
}
int count_map_pairs(Ctx* ctx, hipStream_t stream) {
  const Plan& P = ctx->plan;
  HIP_CHECK(hipMemsetAsync(ctx->dev_pairs, 0, sizeof(unsigned long long) * 8, stream));
  if (P.lv[0].n > 0 && P.g0 && P.t2m)
    hipLaunchKernelGGL(count_k5_pairs_kernel, dim3(512), dim3(256), 0, stream, P.lv[0].keys, P.g0, P.t2m, (int32_t)P.lv[0].n,
                       ctx->dev_pairs);
  for (int l = 1; l < EGONN_NUM_LEVELS; ++l) {
    const int64_t n = P.lv[l].n * 27;
    if (n == 0) continue;
    hipLaunchKernelGGL(count_valid_kernel, dim3(256), dim3(256), 0, stream, P.lv[l].nbr27, n, ctx->dev_pairs + l);
  }
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------ k=2,s=2 tables
__global__ void nbr8_kernel(const int32_t* __restrict__ cstart, const uint64_t* __restrict__ ckeys, int32_t n,
                            int32_t* __restrict__ nbr8) {
  const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  int32_t r[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
  const int32_t s = cstart[p], e = cstart[p + 1];
  for (int32_t c = s; c < e; ++c) {
    const int slot = (int)(ckeys[c] & 7);
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (q == slot) r[q] = c;
  }
  int4* o = reinterpret_cast<int4*>(nbr8 + (int64_t)p * 8);
  o[0] = make_int4(r[0], r[1], r[2], r[3]);
  o[1] = make_int4(r[4], r[5], r[6], r[7]);
}

__global__ void nbrT_kernel(const int32_t* __restrict__ parent, const uint64_t* __restrict__ keys,
                            const int32_t* __restrict__ counts, int32_t cap, int32_t* __restrict__ nbrT) {
  const int32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= level_rows(counts, 0, cap)) return;
  const int slot = (int)(keys[c] & 7);
  const int32_t p = parent[c];
  int32_t r[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) r[q] = (q == slot) ? p : -1;
  int4* o = reinterpret_cast<int4*>(nbrT + (int64_t)c * 8);
  o[0] = make_int4(r[0], r[1], r[2], r[3]);
  o[1] = make_int4(r[4], r[5], r[6], r[7]);
}

// both tables of every level 1..7 in ONE launch (the 14 per-level launches were ~5 us of dispatch each)
struct Nbr8TArgs {
  const int32_t* cstart[EGONN_NUM_LEVELS];
  const uint64_t* ckeys[EGONN_NUM_LEVELS];    // keys of level l-1
  const int32_t* parent[EGONN_NUM_LEVELS];
  const uint64_t* keys[EGONN_NUM_LEVELS];
  int32_t* nbr8[EGONN_NUM_LEVELS];
  int32_t* nbrT[EGONN_NUM_LEVELS];
  const int32_t* counts;                      // device row counts per level
  int32_t prefix[EGONN_NUM_LEVELS + 1];       // prefix over levels 1..7 of their row capacities; [0] unused
};
__device__ static inline void nbr8T_all_body(const Nbr8TArgs& a, uint32_t vblock) {
  const int64_t t = (int64_t)vblock * 256 + threadIdx.x;
  if (t >= a.prefix[EGONN_NUM_LEVELS]) return;
  int l = 1;
  while (l < EGONN_NUM_LEVELS - 1 && t >= a.prefix[l + 1]) ++l;
  const int32_t p = (int32_t)(t - a.prefix[l]);
  if (p >= level_rows(a.counts, l, a.prefix[l + 1] - a.prefix[l])) return;
  {
    int32_t r[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    const int32_t s = a.cstart[l][p], e = a.cstart[l][p + 1];
    const uint64_t* ck = a.ckeys[l];
    for (int32_t c = s; c < e; ++c) {
      const int slot = (int)(ck[c] & 7);
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (q == slot) r[q] = c;
    }
    int4* o = reinterpret_cast<int4*>(a.nbr8[l] + (int64_t)p * 8);
    o[0] = make_int4(r[0], r[1], r[2], r[3]);
    o[1] = make_int4(r[4], r[5], r[6], r[7]);
  }
  {
    const int slot = (int)(a.keys[l][p] & 7);
    const int32_t par = a.parent[l][p];
    int32_t r[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) r[q] = (q == slot) ? par : -1;
    int4* o = reinterpret_cast<int4*>(a.nbrT[l] + (int64_t)p * 8);
    o[0] = make_int4(r[0], r[1], r[2], r[3]);
    o[1] = make_int4(r[4], r[5], r[6], r[7]);
  }
}

// Everything a plan derives from the pyramid alone, in ONE launch (workgroup ranges): the 4x4x4 occupancy masks of levels 2..9,
// the searched adjacency of the two virtual levels, the level-2 block of every level-0 row (first-layer helper) and the k=2 /
// transposed tables of levels 1..7.  Round 5: these were four launches of 4-10 us each in the plan's dependent chain.
struct PlanTablesArgs {
  MaskArgs ma;
  Adj27Args adj;
  Adj27Args adj2;            // levels 6-7 by search as well (a few thousand rows: one nbr27 launch of the chain less)
  Nbr8TArgs na;
  const int32_t *parent0, *parent1;     // grandparent: g0[i] = parent1[parent0[i]]
  int32_t* g0;
  int32_t cap0;
  uint32_t b_mask, b_adj, b_adj2, b_gp, b_n8;   // workgroups of every part, in this order
};
__global__ __launch_bounds__(256) void plan_tables_kernel(const PlanTablesArgs a) {
  uint32_t b = blockIdx.x;
  if (b < a.b_mask) { block_mask_body(a.ma, b); return; }
  b -= a.b_mask;
  if (b < a.b_adj) { adj27_search_body(a.adj, b); return; }
  b -= a.b_adj;
  if (b < a.b_adj2) { adj27_search_body(a.adj2, b); return; }
  b -= a.b_adj2;
  if (b < a.b_gp) {
    const int32_t i = (int32_t)(b * 256 + threadIdx.x);
    if (i < level_rows(a.ma.counts, 0, a.cap0)) a.g0[i] = a.parent1[a.parent0[i]];
    return;
  }
  b -= a.b_gp;
  nbr8T_all_body(a.na, b);
}

__global__ void decode_coords_kernel(const uint64_t* __restrict__ keys, int32_t n, int level, int cb,
                                     int32_t* __restrict__ out) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int cbL = cb - level;
  const uint64_t key = keys[i];
  const uint64_t mort = key & ((1ull << (3 * cbL)) - 1);
  const int32_t bias = 1 << (cb - 1);
  int4 o;
  o.x = (int32_t)(key >> (3 * cbL));
  o.y = ((int32_t)compact1by2(mort) << level) - bias;
  o.z = ((int32_t)compact1by2(mort >> 1) << level) - bias;
  o.w = ((int32_t)compact1by2(mort >> 2) << level) - bias;
  reinterpret_cast<int4*>(out)[i] = o;
}

// ------------------------------------------------------------------ host side
static int batch_bits(int B) {
  int b = 0;
  while ((1 << b) < B) ++b;
  return b < 1 ? 1 : b;
}

// Copies the per-level row counts, the range/overflow flags and the per-sample offsets to the host (the plan's only
// host synchronisation).  Eager plans call it at build time; reserved (capturable) plans only when the host asks for a
// size (egonn_level_count, output allocation) or for the error state.
int plan_sync(Ctx* ctx, hipStream_t stream) {
  Plan& P = ctx->plan;
  EGONN_REQUIRE(P.valid, EGONN_ERR_STATE, "no coordinate plan");
  if (P.exact) return EGONN_OK;
  const int B = P.batch, cb = P.coord_bits;
  HIP_CHECK(hipMemcpyAsync(ctx->host_counts, ctx->dev_counts, sizeof(int32_t) * 17, hipMemcpyDeviceToHost, stream));
  HIP_CHECK(hipMemcpyAsync(ctx->host_counts + 32, P.lv[0].boff, sizeof(int32_t) * EGONN_NUM_LEVELS * (B + 1),
                           hipMemcpyDeviceToHost, stream));
  HIP_CHECK(hipStreamSynchronize(stream));
  const int32_t flags = ctx->host_counts[16];
  if (flags & 1) {
    P.valid = false;
    set_error("coordinate outside the +-2^%d voxel range of coord_bits=%d (or non-finite point / batch index out of range)",
              cb - 1, cb);
    return EGONN_ERR_RANGE;
  }
  if (ctx->host_counts[NL] > B) {
    P.valid = false;
    set_error("batch index %d >= batch size %d", ctx->host_counts[NL] - 1, B);
    return EGONN_ERR_RANGE;
  }
  if (flags & 2) {
    P.valid = false;
    set_error("the batch does not fit the reserved plan (points %d / capacity %lld; level rows %d %d %d %d %d %d %d %d / "
              "capacities %lld %lld %lld %lld %lld %lld %lld %lld): call egonn_ctx_reserve with larger capacities",
              ctx->host_counts[NL + 1], (long long)P.cap_points, ctx->host_counts[0], ctx->host_counts[1], ctx->host_counts[2],
              ctx->host_counts[3], ctx->host_counts[4], ctx->host_counts[5], ctx->host_counts[6], ctx->host_counts[7],
              (long long)P.cap[0], (long long)P.cap[1], (long long)P.cap[2], (long long)P.cap[3], (long long)P.cap[4],
              (long long)P.cap[5], (long long)P.cap[6], (long long)P.cap[7]);
    return EGONN_ERR_CAPACITY;
  }
  for (int l = 0; l < NL; ++l) P.lv[l].n = ctx->host_counts[l];
  for (int l = 0; l < EGONN_NUM_LEVELS; ++l) {
    const int32_t* src = ctx->host_counts + 32 + (size_t)l * (B + 1);
    P.boff_host[l].assign(src, src + B + 1);
  }
  P.n_input = ctx->host_counts[NL + 1];
  P.exact = true;
  if (flags & 8) {             // (the plan itself is fine: the sizes above are valid; the feature maps of the forward are not)
    set_error("an fp32 sparse convolution on the fp16-split matrix pipe met a non-finite accumulator: an activation beyond +-65504 "
              "(the range of the fp16 operand parts) or a non-finite input; the outputs of this batch are invalid — "
              "egonn_ctx_set_exact_fp32(ctx, 1) selects the exact fp32 kernels");
    return EGONN_ERR_FP16_RANGE;
  }
  return EGONN_OK;
}

// n_cap: rows the key buffers hold; n_dev (nullable): device-resident row count.  reserved = false: the level capacities
// are the exact row counts (one host sync right after the pyramid); true: ctx->reserve_cap[] (no host sync at all).
// seg_off (nullable): DEVICE scan offsets (B+1) when the rows arrive scan by scan (plans built from points)
// measurement switch: the round-3 flat sort of (batch | Morton) keys also for plans built from points
static bool plan_flat_sort() {
  static const bool v = getenv("EGONN_FLAT_SORT") != nullptr;
  return v;
}
// bits of the point index inside its scan that a packed sort element can carry next to the 3 * cb Morton bits (0 = pairs)
static int plan_packed_idx_bits(int cb, int64_t n) {
  static const bool off = getenv("EGONN_SORT_PAIRS") != nullptr;          // measurement switch: (key, value) pairs as in round 4
  const int room = std::min(64 - 3 * cb, 30);
  if (off || plan_flat_sort() || room < 1 || n >= (int64_t(1) << room)) return 0;
  return room;
}

static int build_plan_from_sorted_input(Ctx* ctx, uint64_t* keys_raw, uint32_t* vals_raw, uint64_t* keys_sorted,
                                        uint32_t* vals_sorted, int64_t n, const int64_t* n_dev, const int64_t* seg_off, int B,
                                        bool reserved, hipStream_t stream, int idx_bits = 0) {
  Plan& P = ctx->plan;
  const int cb = ctx->coord_bits;
  Arena& A = ctx->plan_arena;
  // (the sorted pairs land in whichever pair the last pass wrote: batches of 17-64 scans need six passes, an even number)
  const bool flat_sort = plan_flat_sort();
  if (seg_off && !flat_sort) {
    // plans built from points: the scans are contiguous, each is sorted on its Morton bits (packed elements when they fit)
    EGONN_TRY(radix_sort_segments(ctx, keys_raw, vals_raw, keys_sorted, vals_sorted, n, seg_off, B, 3 * cb, stream, &keys_sorted,
                                  &vals_sorted, idx_bits));
  } else {
    EGONN_TRY(radix_sort_pairs(ctx, keys_raw, vals_raw, keys_sorted, vals_sorted, n, 3 * cb + batch_bits(B), stream, n_dev, &keys_sorted,
                               &vals_sorted));
  }

  const int ntiles = (int)cdiv(n, PYR_TILE);
  int32_t* tilecnt = A.alloc<int32_t>((size_t)ntiles * NL);
  PyramidOut po;
  for (int l = 0; l < NL; ++l) {
    // rows at level l <= n ; a level-l row needs >= 1 row below, so upper levels could be bounded tighter,
    // but HBM is plentiful: worst case everywhere, the arrays are tiny next to the features.
    P.lv[l] = Level();
    P.lv[l].keys = po.keys[l] = A.alloc<uint64_t>(n);
    P.lv[l].parent = po.parent[l] = A.alloc<int32_t>(n);
    P.lv[l].cstart = po.cstart[l] = A.alloc<int32_t>(n + 1);
    EGONN_REQUIRE(po.keys[l] && po.parent[l] && po.cstart[l], EGONN_ERR_STATE, "plan arena too small");
    // capacity of the level: everything downstream (maps, workspaces, grids) is sized for it
    P.cap[l] = reserved ? std::min<int64_t>(ctx->reserve_cap[l], n) : n;
    po.cap[l] = (int32_t)P.cap[l];
  }
  int32_t* boff_all = A.alloc<int32_t>((size_t)EGONN_NUM_LEVELS * (B + 1));   // contiguous: one D2H copy
  EGONN_REQUIRE(boff_all, EGONN_ERR_STATE, "plan arena too small");
  for (int l = 0; l < EGONN_NUM_LEVELS; ++l) P.lv[l].boff = po.boff[l] = boff_all + (size_t)l * (B + 1);
  P.perm0 = po.perm0 = A.alloc<int32_t>(n);
  po.counts = ctx->dev_counts;
  po.flags = ctx->dev_flags;
  EGONN_REQUIRE(tilecnt && po.perm0, EGONN_ERR_STATE, "plan arena too small");

  hipLaunchKernelGGL(pyramid_count_kernel, dim3(ntiles), dim3(PYR_BLOCK), 0, stream, keys_sorted, n, n_dev, tilecnt);
  hipLaunchKernelGGL(pyramid_apply_kernel, dim3(ntiles), dim3(PYR_BLOCK), 0, stream, keys_sorted, vals_sorted, n, n_dev,
                     tilecnt, ntiles, cb, B, po);
  HIP_CHECK(hipGetLastError());
  P.batch = B;
  P.coord_bits = cb;
  P.n_input = n;
  P.cap_points = n;
  P.valid = true;
  P.exact = false;
  P.built_reserved = reserved;
  if (!reserved) {
    // ---- the size query (single host sync of an eager plan): counts, range flag, per-sample offsets
    EGONN_TRY(plan_sync(ctx, stream));
    for (int l = 0; l < NL; ++l) P.cap[l] = P.lv[l].n;
  } else {
    for (int l = 0; l < NL; ++l) P.lv[l].n = P.cap[l];      // upper bounds until somebody asks (plan_sync)
  }
  P.valid = false;                                           // until the maps are enqueued
  const int32_t* counts = ctx->dev_counts;

  // ---- one launch for everything that needs the pyramid only (plan_tables_kernel)
  PlanTablesArgs ta;
  MaskArgs& ma = ta.ma;
  int32_t pre = 0;
  for (int L = 0; L < NL; ++L) {
    ma.cstartL[L] = P.lv[L].cstart;
    ma.keys[L] = P.lv[L].keys;
    ma.mask[L] = nullptr;
    ma.bstart[L] = nullptr;
    ma.prefix[L] = 0;
  }
  ma.counts = counts;
  for (int L = 2; L < NL; ++L) {
    P.lv[L].mask = ma.mask[L] = A.alloc<uint64_t>(P.cap[L]);
    P.lv[L].bstart = ma.bstart[L] = A.alloc<int32_t>(P.cap[L]);
    EGONN_REQUIRE(ma.mask[L] && ma.bstart[L], EGONN_ERR_STATE, "plan arena too small");
    ma.prefix[L] = pre;
    pre += (int32_t)P.cap[L];
  }
  ma.prefix[NL] = pre;
  ta.b_mask = (unsigned)cdiv(pre, 256);
  // the k=3 tables of every level (allocated top-down as before)
  for (int l = NL - 1; l >= 1; --l) {
    P.lv[l].nbr27 = A.alloc<int32_t>((size_t)P.cap[l] * 27);
    EGONN_REQUIRE(P.lv[l].nbr27, EGONN_ERR_STATE, "plan arena too small");
  }
  {   // the two virtual levels by search (tiny)
    Level& V = P.lv[NL - 2];
    Level& V1 = P.lv[NL - 1];
    const int32_t nv = (int32_t)P.cap[NL - 2], nv1 = (int32_t)P.cap[NL - 1];
    ta.adj = Adj27Args{V.keys, V1.keys, counts, NL - 2, NL - 1, cb - (NL - 2), cb - (NL - 1), nv, nv1, V.nbr27, V1.nbr27};
    ta.b_adj = (unsigned)cdiv(((int64_t)nv + nv1) * 27, 256);
  }
  // levels 6 and 7 (1 488 + 682 rows at batch 16) by search too: bitwise the tables nbr27_kernel derives from levels 8 / 9, and the
  // chain of dependent nbr27 launches is three instead of four.  (Capped: a level beyond 8 192 rows keeps the derived path.)
  static const bool search67_ok = getenv("EGONN_NO_SEARCH67") == nullptr;              // measurement switch
  const int search_from = (search67_ok && P.cap[NL - 4] <= 8192 && P.cap[NL - 3] <= 8192) ? NL - 4 : NL - 2;
  ta.b_adj2 = 0;
  ta.adj2 = ta.adj;
  if (search_from == NL - 4) {
    Level& V = P.lv[NL - 4];
    Level& V1 = P.lv[NL - 3];
    const int32_t nv = (int32_t)P.cap[NL - 4], nv1 = (int32_t)P.cap[NL - 3];
    ta.adj2 = Adj27Args{V.keys, V1.keys, counts, NL - 4, NL - 3, cb - (NL - 4), cb - (NL - 3), nv, nv1, V.nbr27, V1.nbr27};
    ta.b_adj2 = (unsigned)cdiv(((int64_t)nv + nv1) * 27, 256);
  }
  const int32_t n0 = (int32_t)P.cap[0], n2 = (int32_t)P.cap[2];
  {   // first-layer (k=5) helpers
    P.g0 = A.alloc<int32_t>(n0);
    P.t2m = A.alloc<uint64_t>((size_t)n2 * 27);
    P.t2s = A.alloc<int32_t>((size_t)n2 * 27);
    EGONN_REQUIRE(P.g0 && P.t2m && P.t2s, EGONN_ERR_STATE, "plan arena too small");
    ta.parent0 = P.lv[0].parent; ta.parent1 = P.lv[1].parent; ta.g0 = P.g0; ta.cap0 = n0;
    ta.b_gp = (unsigned)cdiv(n0, 256);
  }
  {
    Nbr8TArgs& na = ta.na;
    na.prefix[0] = na.prefix[1] = 0;
    na.counts = counts;
    for (int l = 1; l < EGONN_NUM_LEVELS; ++l) {
      Level& V = P.lv[l];
      const int32_t nv = (int32_t)P.cap[l];
      V.nbr8 = A.alloc<int32_t>((size_t)nv * 8);
      V.nbrT = A.alloc<int32_t>((size_t)nv * 8);
      EGONN_REQUIRE(V.nbr8 && V.nbrT, EGONN_ERR_STATE, "plan arena too small");
      na.cstart[l] = V.cstart;
      na.ckeys[l] = P.lv[l - 1].keys;
      na.parent[l] = V.parent;
      na.keys[l] = V.keys;
      na.nbr8[l] = V.nbr8;
      na.nbrT[l] = V.nbrT;
      na.prefix[l + 1] = na.prefix[l] + nv;
    }
    na.cstart[0] = nullptr; na.ckeys[0] = nullptr; na.parent[0] = nullptr; na.keys[0] = nullptr;
    na.nbr8[0] = nullptr; na.nbrT[0] = nullptr;
    ta.b_n8 = (unsigned)cdiv(na.prefix[EGONN_NUM_LEVELS], 256);
  }
  {
    const unsigned nb = ta.b_mask + ta.b_adj + ta.b_adj2 + ta.b_gp + ta.b_n8;
    if (nb > 0) hipLaunchKernelGGL(plan_tables_kernel, dim3(nb), dim3(256), 0, stream, ta);
  }

  // ---- k=3 kernel maps, top-down: every level from the k=3 table of the level two above (= the adjacency of its 4x4x4
  //      blocks); levels (l, l-1) are independent and share a launch.  The last launch (level 1) also carries the first
  //      layer's per-block neighbour table (blk27: reads the level-2 table, finished by the launch before).
  Nbr27Pair pair;
  int npair = 0;
  pair.counts = counts;
  pair.split = 0;
  pair.tail_at = ~0u;
  pair.blk = Blk27Args{nullptr, nullptr, nullptr, 0, nullptr, nullptr};
  bool blk_done = false;
  auto flush = [&](bool last) {
    if (npair == 0 && !(last && !blk_done && n0 > 0)) return;
    if (npair == 1) pair.job[1] = pair.job[0];
    const unsigned g0 = npair >= 1 ? pair.split : 0u, g1 = npair == 2 ? (unsigned)cdiv(pair.job[1].cap_blocks, 4) : 0u;
    if (npair == 0) { pair.split = 0; pair.job[0] = Nbr27Job{nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0, 0}; pair.job[1] = pair.job[0]; }
    unsigned extra = 0;
    pair.tail_at = ~0u;
    if (last && !blk_done && n0 > 0) {
      pair.tail_at = g0 + g1;
      pair.blk = Blk27Args{P.lv[2].nbr27, P.lv[2].mask, P.lv[2].bstart, n2, P.t2m, P.t2s};
      extra = (unsigned)cdiv((int64_t)n2 * 27, 256);
      blk_done = true;
    }
    if (g0 + g1 + extra > 0) hipLaunchKernelGGL(nbr27_kernel, dim3(g0 + g1 + extra), dim3(256), 0, stream, pair);
    npair = 0;
  };
  for (int l = search_from - 1; l >= 1; --l) {
    Level& V = P.lv[l];
    const int32_t nv = (int32_t)P.cap[l];
    if (nv == 0) continue;
    const Level& Bk = P.lv[l + 2];
    // levels (NL-3, NL-4), (NL-5, NL-6), ... pair up: both members of a pair read tables finished by earlier launches
    if (npair == 1 && pair.job[0].level != l + 1) flush(false);
    Nbr27Job& J = pair.job[npair];
    J.vkeys = V.keys; J.badj = Bk.nbr27; J.bmask = Bk.mask; J.bstart = Bk.bstart; J.nbr = V.nbr27;
    J.level = l; J.cap_blocks = (int32_t)P.cap[l + 2]; J.cap_vox = nv;
    if (npair == 0) pair.split = (unsigned)cdiv(P.cap[l + 2], 4);
    ++npair;
    if (npair == 2) flush(false);
  }
  flush(true);
  HIP_CHECK(hipGetLastError());
  P.valid = true;
  return EGONN_OK;
}

// level-0 parent table (input gradient of the first strided convolution): built on first use, training only
int ensure_level0_parent_table(Ctx* ctx, hipStream_t stream) {
  Plan& P = ctx->plan;
  Level& V = P.lv[0];
  if (V.nbrT || P.cap[0] == 0) return EGONN_OK;
  V.nbrT = ctx->plan_arena.alloc<int32_t>((size_t)P.cap[0] * 8);
  EGONN_REQUIRE(V.nbrT, EGONN_ERR_STATE, "plan arena too small for the level-0 parent table");
  hipLaunchKernelGGL(nbrT_kernel, dim3((unsigned)cdiv(P.cap[0], 256)), dim3(256), 0, stream, V.parent, V.keys,
                     ctx->dev_counts, (int32_t)P.cap[0], V.nbrT);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------ row-group tables (rowgroup.hip) of the plan's maps
static int rg_window(int level) { return level <= 3 ? 512 : 256; }   // measured: 256 everywhere costs 7 % on the level-1 convs (builder no faster); 512 up to level 5 costs 5 % on the 128-channel convs

int rowgroup_cap_groups(const Plan& P, int level) {
  const int win = rg_window(level);
  return (int)((cdiv(P.cap[level], win) + P.batch) * (win / 16));
}

// Builds the row-group form of the requested maps that do not exist yet, all in ONE launch.
// kind: 0 = k=3 map of `level`, 1 = k=2,s=2 map into `level` (from level-1), 2 = transposed map onto `level` (from level+1)
int ensure_rowgroups(Ctx* ctx, const int* kinds, const int* levels, int count, hipStream_t stream) {
  Plan& P = ctx->plan;
  RGBuild jobs[RG_MAX_JOBS];
  int nj = 0;
  for (int i = 0; i < count; ++i) {
    const int kind = kinds[i], l = levels[i];
    EGONN_REQUIRE(kind >= 0 && kind <= 2 && l >= (kind == 2 ? 0 : 1) && l < EGONN_NUM_LEVELS - (kind == 2 ? 1 : 0),
                  EGONN_ERR_INVALID, "rowgroups: map kind %d / level %d out of range", kind, l);
    Level& V = P.lv[l];
    RowGroups& rg = kind == 0 ? V.rg27 : (kind == 1 ? V.rg8 : V.rgT);
    if (rg.built) continue;
    bool dup = false;
    for (int j = 0; j < nj; ++j) dup |= (jobs[j].rg == &rg);
    if (dup) continue;
    if (kind == 2 && l == 0) EGONN_TRY(ensure_level0_parent_table(ctx, stream));
    const int32_t* nbr = kind == 0 ? V.nbr27 : (kind == 1 ? V.nbr8 : V.nbrT);
    EGONN_REQUIRE(nbr, EGONN_ERR_STATE, "rowgroups: the plan has no kernel map of kind %d at level %d", kind, l);
    rg.K = kind == 0 ? 27 : 8;
    rg.win = rg_window(l);
    rg.cap_groups = rowgroup_cap_groups(P, l);
    Arena& A = ctx->plan_arena;
    rg.perm = A.alloc<int32_t>((size_t)rg.cap_groups * 16);
    rg.snbr = A.alloc<int32_t>((size_t)rg.cap_groups * rg.K * 16);
    rg.gmask = A.alloc<uint32_t>((size_t)rg.cap_groups);
    rg.meta = A.alloc<int32_t>((size_t)P.batch + 2);
    rg.order4 = A.alloc<int32_t>((size_t)rg.cap_groups / 4 + 16);
    EGONN_REQUIRE(rg.perm && rg.snbr && rg.gmask && rg.meta && rg.order4, EGONN_ERR_STATE, "plan arena too small (row groups)");
    EGONN_REQUIRE(nj < RG_MAX_JOBS, EGONN_ERR_INVALID, "rowgroups: too many maps in one request");
    jobs[nj].rg = &rg;
    jobs[nj].nbr = nbr;
    jobs[nj].n_dev = ctx->dev_counts + l;
    jobs[nj].boff = V.boff;
    jobs[nj].cap_rows = (int32_t)P.cap[l];
    ++nj;
  }
  if (nj == 0) return EGONN_OK;
  EGONN_TRY(rowgroup_build(jobs, nj, P.batch, stream));
  for (int j = 0; j < nj; ++j) jobs[j].rg->built = true;
  return EGONN_OK;
}

static size_t plan_arena_bytes(int64_t n, int B) {
  // raw+sorted keys/vals, 10 levels x (keys, parent, cstart, mask, bstart), perm, maps of levels 1..7 (<= n rows each)
  size_t per_row = 2 * (8 + 4) + NL * (8 + 4 + 4 + 8 + 4) + 4 + 9 * 27 * 4 + 7 * (8 + 8) * 4 + 4 + 27 * 12 + 8 * 4;
  // row-group tables: k=3 (27+1 ints + mask) and the two 8-slot maps, <= 2n rows over all levels + window rounding
  per_row += 2 * ((27 + 1) * 4 + 2 + 2 * ((8 + 1) * 4 + 2));
  const size_t rg_round = (size_t)(B + 8) * 1024 * (28 + 14 + 2 * 9) * 4 * EGONN_NUM_LEVELS;
  return (size_t)(n + 8) * per_row + (size_t)(B + 1) * 4 * EGONN_NUM_LEVELS + (size_t)cdiv(n, PYR_TILE) * NL * 4 + rg_round +
         (1 << 20);
}

// scan_offsets: HOST (B+1) when offsets_on_device == 0 (eager plan: exact capacities, one host sync), DEVICE int64 (B+1)
// otherwise (reserved plan: capacities from egonn_ctx_reserve, no host sync, capturable; n_cap = rows of `points`).
int plan_from_points(Ctx* ctx, const float* points, const int64_t* scan_offsets, int64_t n_cap, int offsets_on_device,
                     int B, int mode, const float* step, hipStream_t stream) {
  ctx->plan.valid = false;
  EGONN_REQUIRE(B >= 1 && B <= EGONN_MAX_BATCH, EGONN_ERR_INVALID, "batch size %d outside [1,%d]", B, EGONN_MAX_BATCH);
  EGONN_REQUIRE(mode == 0 || mode == 1, EGONN_ERR_INVALID, "unknown quantiser mode %d", mode);
  int64_t n;
  if (!offsets_on_device) {
    n = scan_offsets[B];
    EGONN_REQUIRE(scan_offsets[0] == 0 && n >= 1, EGONN_ERR_INVALID, "empty input (n=%lld points)", (long long)n);
    for (int b = 0; b < B; ++b)
      EGONN_REQUIRE(scan_offsets[b] <= scan_offsets[b + 1], EGONN_ERR_INVALID, "scan offsets not monotone");
    EGONN_TRY(ctx->plan_arena.ensure(plan_arena_bytes(n, B)));
  } else {
    n = n_cap;
    EGONN_REQUIRE(ctx->reserved && n >= 1 && n <= ctx->reserve_points && B == ctx->reserve_batch, EGONN_ERR_STATE,
                  "device-offset plans need egonn_ctx_reserve (points %lld / reserved %lld, batch %d / %d)", (long long)n,
                  (long long)ctx->reserve_points, B, ctx->reserve_batch);
  }
  Arena& A = ctx->plan_arena;
  A.reset();
  uint64_t* k0 = A.alloc<uint64_t>(n);
  uint64_t* k1 = A.alloc<uint64_t>(n);
  uint32_t* v0 = A.alloc<uint32_t>(n);
  uint32_t* v1 = A.alloc<uint32_t>(n);
  int64_t* doff = A.alloc<int64_t>(B + 1);
  EGONN_REQUIRE(k0 && k1 && v0 && v1 && doff, EGONN_ERR_STATE, "plan arena too small");
  if (!offsets_on_device) {
    // pageable caller memory -> the context's pinned staging buffer -> device (the caller may reuse its array at once)
    int64_t* stage = reinterpret_cast<int64_t*>(ctx->host_counts + 32 + (size_t)EGONN_NUM_LEVELS * (EGONN_MAX_BATCH + 1));
    memcpy(stage, scan_offsets, sizeof(int64_t) * (B + 1));
    HIP_CHECK(hipMemcpyAsync(doff, stage, sizeof(int64_t) * (B + 1), hipMemcpyHostToDevice, stream));
  }
  // (device offsets: the key kernel copies them into the plan's array itself)
  ctx->plan.scan_off = doff;
  HIP_CHECK(hipMemsetAsync(ctx->dev_flags, 0, sizeof(int32_t), stream));
  QuantParams qp{mode, step[0], mode ? step[1] : step[0], mode ? step[2] : step[0]};
  const int idx_bits = plan_packed_idx_bits(ctx->coord_bits, n);
  int32_t *tilehist = nullptr, *scanhist = nullptr;
  int zero_words = 0;
  ctx->sort_prezeroed = nullptr;
  if (!plan_flat_sort()) {       // the segmented sort follows: its per-scan histograms are zeroed by the key kernel
    EGONN_TRY(radix_sort_segments_layout(ctx, n, B, &tilehist, &scanhist));
    zero_words = radix_sort_segments_passes(3 * ctx->coord_bits) * B * radix_sort_segments_digits();
    ctx->sort_prezeroed = scanhist;
  }
  hipLaunchKernelGGL(points_to_keys_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, points, n,
                     offsets_on_device ? scan_offsets : doff, B, qp, ctx->coord_bits, k0, v0, ctx->dev_flags, idx_bits,
                     offsets_on_device ? doff : (int64_t*)nullptr, scanhist, zero_words);
  return build_plan_from_sorted_input(ctx, k0, v0, k1, v1, n, doff + B, doff, B, offsets_on_device != 0, stream, idx_bits);
}

int plan_from_coords(Ctx* ctx, const int32_t* coords, int64_t n, int B, hipStream_t stream) {
  ctx->plan.valid = false;
  EGONN_REQUIRE(n >= 1, EGONN_ERR_INVALID, "empty coordinate list");
  EGONN_REQUIRE(B >= 1 && B <= EGONN_MAX_BATCH, EGONN_ERR_INVALID, "batch size %d outside [1,%d]", B, EGONN_MAX_BATCH);
  EGONN_TRY(ctx->plan_arena.ensure(plan_arena_bytes(n, B)));
  Arena& A = ctx->plan_arena;
  A.reset();
  uint64_t* k0 = A.alloc<uint64_t>(n);
  uint64_t* k1 = A.alloc<uint64_t>(n);
  uint32_t* v0 = A.alloc<uint32_t>(n);
  uint32_t* v1 = A.alloc<uint32_t>(n);
  EGONN_REQUIRE(k0 && k1 && v0 && v1, EGONN_ERR_STATE, "plan arena too small");
  ctx->plan.scan_off = nullptr;
  HIP_CHECK(hipMemsetAsync(ctx->dev_flags, 0, sizeof(int32_t), stream));
  hipLaunchKernelGGL(coords_to_keys_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, coords, n,
                     ctx->coord_bits, B, k0, v0, ctx->dev_flags);
  return build_plan_from_sorted_input(ctx, k0, v0, k1, v1, n, nullptr, nullptr, B, false, stream);
}

// Fixes the sizes of everything a plan allocates, so that later plans neither allocate nor synchronise (capturable).
int plan_reserve(Ctx* ctx, int64_t max_points, int B, const int64_t* level_caps) {
  EGONN_REQUIRE(max_points >= 1 && B >= 1 && B <= EGONN_MAX_BATCH, EGONN_ERR_INVALID, "reserve: bad sizes");
  ctx->plan.valid = false;
  for (int l = 0; l < NL; ++l) {
    int64_t c = max_points;
    if (level_caps && l < EGONN_NUM_LEVELS) c = std::min<int64_t>(std::max<int64_t>(level_caps[l], 64), max_points);
    if (level_caps && l >= EGONN_NUM_LEVELS) c = ctx->reserve_cap[EGONN_NUM_LEVELS - 1];   // virtual levels: <= level 7
    ctx->reserve_cap[l] = c;
  }
  ctx->reserve_points = max_points;
  ctx->reserve_batch = B;
  EGONN_TRY(ctx->plan_arena.ensure(plan_arena_bytes(max_points, B)));
  EGONN_TRY(ctx->sort_arena.ensure(std::max(radix_sort_scratch_bytes(max_points), radix_sort_segments_scratch_bytes(max_points, B))));
  ctx->reserved = true;
  return EGONN_OK;
}

int plan_level_coords(Ctx* ctx, int level, int32_t* out, hipStream_t stream) {
  EGONN_REQUIRE(ctx->plan.valid, EGONN_ERR_STATE, "no coordinate plan (call egonn_voxelize / egonn_coords_set first)");
  EGONN_REQUIRE(level >= 0 && level < EGONN_MAX_LEVELS, EGONN_ERR_INVALID, "level %d out of range", level);
  EGONN_TRY(plan_sync(ctx, stream));
  const Level& L = ctx->plan.lv[level];
  if (L.n == 0) return EGONN_OK;
  hipLaunchKernelGGL(decode_coords_kernel, dim3((unsigned)cdiv(L.n, 256)), dim3(256), 0, stream, L.keys, (int32_t)L.n,
                     level, ctx->plan.coord_bits, out);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

}  // namespace egonn
