// Shared host/device definitions for libegonn_hip (gfx950 / MI355X only).
//
// Data model:
//   * every voxel of every scan in a batch is one 64-bit Z-order key
//         key = (batch << 3*CB) | morton3(x + 2^(CB-1), y + 2^(CB-1), z + 2^(CB-1))
//     with x in bit 0, y in bit 1, z in bit 2 of every triple and CB = coord_bits (10..16).
//   * rows of every level are stored sorted by key  =>  rows are batch-contiguous, the 8
//     children of a stride-2 parent are adjacent and `key >> 3` is the parent's key, and
//     `key & 7` is MinkowskiEngine's kernel index of a k=2,s=2 convolution (x fastest).
//   * level l keys are level-0 keys >> 3l; decoded coordinates are multiples of 2^l, equal to
//     ME's floor(c / 2^l) * 2^l because the bias 2^(CB-1) is a multiple of 2^9.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#define EGONN_MAX_LEVELS 10      // levels 0..9 (7 real strided levels + 2 virtual ones for block masks)
#define EGONN_NUM_LEVELS 8       // levels that carry features: 0..7
#define EGONN_MAX_BATCH 4096

namespace egonn {

// ------------------------------------------------------------------ error plumbing
void set_error(const char* fmt, ...);
#define EGONN_OK 0
#define EGONN_ERR_INVALID 1
#define EGONN_ERR_HIP 2
#define EGONN_ERR_RANGE 3
#define EGONN_ERR_STATE 4
#define EGONN_ERR_CAPACITY 5   // a batch did not fit the capacities of egonn_ctx_reserve (the eager path still works)
#define EGONN_ERR_FP16_RANGE 6 // an fp32 sparse convolution on the fp16-split pipe met a non-finite accumulator (activation beyond
                               // +-65504 or non-finite input): rerun with egonn_ctx_set_exact_fp32(ctx, 1)

#define HIP_CHECK(expr)                                                                   \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      ::egonn::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,               \
                         hipGetErrorString(_e));                                          \
      return EGONN_ERR_HIP;                                                               \
    }                                                                                     \
  } while (0)

#define EGONN_REQUIRE(cond, code, ...)                                                    \
  do {                                                                                    \
    if (!(cond)) {                                                                        \
      ::egonn::set_error(__VA_ARGS__);                                                    \
      return (code);                                                                      \
    }                                                                                     \
  } while (0)

#define EGONN_TRY(expr)                                                                   \
  do {                                                                                    \
    int _rc = (expr);                                                                     \
    if (_rc != EGONN_OK) return _rc;                                                      \
  } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// "set this kernel's dynamic-LDS attribute once" — once per (kernel, DEVICE), safe from several host threads (a process may
// hold contexts on several GPUs; a racing second call only repeats an idempotent setting)
struct AttrOnce {
  std::atomic<uint64_t> done{0};
  static int dev() { int d = 0; (void)hipGetDevice(&d); return d & 63; }
  bool need() const { return !((done.load(std::memory_order_acquire) >> dev()) & 1ull); }
  void mark() { done.fetch_or(1ull << dev(), std::memory_order_release); }
};
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ------------------------------------------------------------------ device arena (grow-only)
struct Arena {
  char* base = nullptr;
  size_t cap = 0;
  size_t off = 0;
  int ensure(size_t bytes);                 // may hipFree + hipMalloc (only legal when nothing is live)
  void reset() { off = 0; }
  template <typename T>
  T* alloc(size_t n) {
    size_t o = align_up(off, 256);
    size_t end = o + n * sizeof(T);
    if (end > cap) return nullptr;
    off = end;
    return reinterpret_cast<T*>(base + o);
  }
  void release();
};

// ------------------------------------------------------------------ Morton helpers (host + device)
__host__ __device__ static inline uint64_t part1by2(uint32_t v) {   // 16 bits -> every 3rd bit
  uint64_t x = v & 0xFFFFull;
  x = (x | (x << 16)) & 0x0000FF0000FFull;
  x = (x | (x << 8)) & 0x00F00F00F00Full;
  x = (x | (x << 4)) & 0x0C30C30C30C3ull;
  x = (x | (x << 2)) & 0x249249249249ull;
  return x;
}
__host__ __device__ static inline uint32_t compact1by2(uint64_t x) {
  x &= 0x249249249249ull;
  x = (x | (x >> 2)) & 0x0C30C30C30C3ull;
  x = (x | (x >> 4)) & 0x00F00F00F00Full;
  x = (x | (x >> 8)) & 0x0000FF0000FFull;
  x = (x | (x >> 16)) & 0xFFFFull;
  return (uint32_t)x;
}
__host__ __device__ static inline uint64_t morton3(uint32_t x, uint32_t y, uint32_t z) {
  return part1by2(x) | (part1by2(y) << 1) | (part1by2(z) << 2);
}

// ------------------------------------------------------------------ row-group tables (rowgroup.hip, sconv.hip)
// The form of a kernel map the sparse-convolution kernel consumes: rows regrouped (per sample, per window, sorted by
// neighbour-presence mask) into groups of 16 that share their set of present kernel offsets.
static constexpr int RG_MAX_JOBS = 24;
static constexpr int RG_MAX_WIN = 512;
struct RowGroups {
  int K = 0;                  // kernel volume of the map (27 or 8)
  int win = 0;                // rows per sort window (256 / 512 / 1024); groups per window = win / 16
  int cap_groups = 0;         // groups the arrays can hold (multiple of win / 16)
  int32_t* perm = nullptr;    // [cap_groups][16]      output row of every slot, -1 = padding
  int32_t* snbr = nullptr;    // [cap_groups][K][16]   input row, -1 = no neighbour
  uint32_t* gmask = nullptr;  // [cap_groups]          OR of the 16 presence masks; bit 31 = group has real rows
  int32_t* meta = nullptr;    // [0] groups in use, [1 + b] first group of sample b (b = 0..B)
  int32_t* order4 = nullptr;  // [cap_groups / 4] tasks of 4 consecutive groups in dispatch order: inside each contiguous eighth
                              // (= XCD) the tasks with the most offsets first (the tail of a launch is then made of short tasks)
  bool built = false;
};
struct RGBuild {
  RowGroups* rg;
  const int32_t* nbr;         // [n][K] kernel map
  const int32_t* n_dev;       // device: rows of the output level
  const int32_t* boff;        // device: [B+1] per-sample offsets of the output level
  int32_t cap_rows = 0;       // rows the level's arrays (and the map) can hold: rows beyond it are never touched
};
int rowgroup_build(const RGBuild* jobs, int njobs, int B, hipStream_t stream);

// ------------------------------------------------------------------ coordinate plan ("coordinate manager")
struct Level {
  int64_t n = 0;              // rows at this level (host copy, valid after the size query)
  uint64_t* keys = nullptr;   // [n]     sorted unique keys (level-l key = level-0 key >> 3l)
  int32_t* parent = nullptr;  // [n]     row of the parent at level l+1
  int32_t* cstart = nullptr;  // [n+1]   first child row at level l-1 (l >= 1)
  uint64_t* mask = nullptr;   // [n]     (l >= 2) 4x4x4 occupancy of the level-(l-2) voxels inside this block
  int32_t* bstart = nullptr;  // [n]     (l >= 2) first level-(l-2) row inside this block
  int32_t* boff = nullptr;    // [B+1]   first row of every sample (levels 0..7)
  int32_t* nbr27 = nullptr;   // [n][27] k=3 neighbour rows at the same level (levels 1..7), -1 = absent
  int32_t* nbr8 = nullptr;    // [n][8]  k=2,s=2 children rows at level l-1 by kernel slot (levels 1..7)
  int32_t* nbrT = nullptr;    // [n][8]  transposed conv: parent row (at level l+1) in slot (key&7), else -1
  RowGroups rg27, rg8, rgT;   // row-group form of nbr27 / nbr8 / nbrT (built on first use: ensure_rowgroups)
};

struct Plan {
  bool valid = false;
  bool built_reserved = false;   // built by egonn_voxelize_device: its structure (pointers, capacities) survives a failed batch
  bool exact = false;         // lv[l].n / boff_host hold the true sizes (false: reserved plan not yet synchronised; n = capacity)
  int64_t cap_points = 0;     // rows of the key buffers
  int64_t* scan_off = nullptr;   // device copy of the scan offsets (voxelize plans)
  int batch = 0;              // B
  int coord_bits = 16;        // CB
  int64_t n_input = 0;        // rows/points handed in by the caller
  Level lv[EGONN_MAX_LEVELS];
  int64_t cap[EGONN_MAX_LEVELS] = {};   // row capacity per level that grids / workspaces are sized for (>= lv[l].n)
  int32_t* perm0 = nullptr;   // [n0] caller row / point index that became sorted row i (first occurrence)
  int32_t* g0 = nullptr;      // [n0] level-2 row (4x4x4 block) of every level-0 row
  uint64_t* t2m = nullptr;    // [n2][27] occupancy masks of the 27 blocks around every level-2 row (0 = absent)
  int32_t* t2s = nullptr;     // [n2][27] first level-0 row of those blocks
  std::vector<int32_t> boff_host[EGONN_NUM_LEVELS];   // host copies of boff (size B+1)
};

// ------------------------------------------------------------------ per-launch HIP-event timing (bench.py roofline leg)
enum ProfKind { PK_CONV0 = 0, PK_K3 = 1, PK_K2S2 = 2, PK_TCONV = 3, PK_OTHER = 4 };
struct ProfRec {
  char name[64];
  int kind, level, K, cin, cout;     // enough to evaluate the algorithmic-bytes formula of SURVEY.md §8(d)
  int es;                            // bytes per feature-map / weight element (4 fp32, 2 bf16)
  hipEvent_t e0, e1;
};
struct Profiler {
  int mode = 0;                       // 0 off, 1 all tagged launches, 2 only launches whose name contains `filter` (exact
                                      // dispatch timing), 3 = filtered launches bracketed by event records (capturable)
  char filter[64] = "";
  std::vector<ProfRec> recs;
  std::vector<ProfRec> graph_recs;    // brackets recorded while a stream capture was active: re-recorded by every replay
  std::vector<hipEvent_t> pool;
  hipEvent_t get();
};

// Offset-split rule of the fp32 lock-step kernels (sconv_ksplit_rule): [0] = k=3 maps, [1] = 8-slot maps, indexed by output level
struct KsRule {
  int8_t kparts[2][EGONN_NUM_LEVELS];   // offset parts as separate workgroups + reducer launch (1 = none)
  int8_t kw[2][EGONN_NUM_LEVELS];       // offset parts inside a workgroup (0 / 1 = none)
  int8_t col_parts[EGONN_NUM_LEVELS];   // column parts per task (0 = automatic)
};

struct Ctx {
  Profiler prof;
  KsRule ks_rule;             // set at context creation (sconv_ksplit_defaults); egonn_debug_set_ksplit
  uint16_t* conv0_lut = nullptr;             // first-layer lookup table (64 positions x 128 offsets), built on first use
  unsigned long long* dev_pairs = nullptr;   // [16] kernel-map pair counters: [0] conv0 k5, [l] k3 map of level l
  int device = 0;
  int coord_bits = 16;
  int split_max_level = 5;    // fp32 sparse convs whose output level is <= this run on the fp16-split kernels (sconv_split.hip);
                              // round 5: 4 -> 5 (profiles/r05b_tail_kernel.txt: 25.2 k -> 26.4 k scans/s with four batches in flight; the
                              // level-5 launches alone get slower, one-batch graph latency 1.17 -> 1.23 ms — the headline metric is scans/s)
  int split_io = 0;           // set by egonn_forward around ONE sconv_map call: bit 0 = the input map is in split form (fp16 hi | lo
                              // per 32-channel block, sconv_split.hip), bit 1 = write the output in split form
  const float* gated_in2 = nullptr;   // set by egonn_forward around ONE sconv_map call: the convolution's input row r is
  const float* gated_gate = nullptr;  // relu(in[r] * gate[scan] + in2[r]) — the tail of the ECA block below, never materialised
  float* ks_part = nullptr;   // scratch for the partial tiles of the offset-split launches (sconv_split.hip): carved from the work arena
  size_t ks_part_floats = 0;  // by egonn_forward / the stand-alone operator entry points (sconv_ksplit_scratch_floats)
  const void* sort_prezeroed = nullptr;   // the per-scan histograms of the segmented sort were zeroed by the kernel in front of it
  int keep_level_features = 0;   // egonn_debug_keep_level_features: no fusion that leaves a level's block output unmaterialised
  const float* conv_residual = nullptr;   // set by egonn_forward around ONE sconv_map call (fp32 maps): out += residual in the epilogue
  int operand_autoscale = 0;  // egonn_ctx_set_operand_autoscale: the fp16-split convolutions scale their INPUT by a power of two per launch
                              // (max |in| -> [2^13, 2^14), undone in the epilogue): the input-gradient convolutions of a training step
  int conv_variant = 0;       // tests / A-B measurements only (egonn_debug_set_naive_conv): 0 = product choice, 1 = per-wave
                              // MFMA kernel, 2 = workgroup-cooperative MFMA kernel, 3 = plain one-thread-per-output kernel
  Arena plan_arena;           // keys, maps (lives until the next plan)
  Arena work_arena;           // features & scratch of one forward
  Arena sort_arena;           // radix-sort scratch
  Plan plan;
  int32_t* host_counts = nullptr;    // pinned staging for the size query
  int32_t* dev_counts = nullptr;
  int32_t* dev_flags = nullptr;      // bit 0 = out-of-range coordinate seen, bit 1 = batch larger than the reserved capacities,
                                     // bit 3 = fp16 range guard of the split convolutions (sconv_split.hip)
  bool reserved = false;             // egonn_ctx_reserve: fixed capacities, plans neither allocate nor synchronise
  int64_t reserve_points = 0;
  int reserve_batch = 0;
  int64_t reserve_cap[EGONN_MAX_LEVELS] = {};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

// RAII launch timer: records events on `stream` around the enclosed launches when profiling is enabled.
// In mode 2 the scope does not record anything itself: it parks its event pair in `prof_kernel_events()` and the
// sparse-conv launcher attaches them to the kernel dispatch (hipExtLaunchKernelGGL start/stop events), so that the
// elapsed time is the kernel's own begin..end — the figure rocprofv3 --kernel-trace reports — even when other
// streams share the GPU.  Mode 1 brackets all launches of the scope with ordinary stream events.
hipEvent_t* prof_kernel_events();     // thread-local [2]; {nullptr, nullptr} when no exact timing is requested
struct ProfScope {
  Ctx* ctx;
  hipStream_t st;
  int idx = -1;
  bool exact = false;
  bool in_graph = false;
  ProfScope(Ctx* c, hipStream_t s, const char* name, int kind, int level, int K, int cin, int cout, int es);
  ~ProfScope();
};

int ensure_level0_parent_table(Ctx* ctx, hipStream_t stream);   // coords.hip
// row-group form of kernel maps (coords.hip): kind 0 = k=3 map of `level`, 1 = k=2,s=2 map into `level`,
// 2 = transposed map onto `level`; every missing table of the request is built in one launch
int ensure_rowgroups(Ctx* ctx, const int* kinds, const int* levels, int count, hipStream_t stream);
int rowgroup_cap_groups(const Plan& P, int level);   // groups the row-group tables of a map onto `level` hold (a function of P.cap, P.batch)

// ------------------------------------------------------------------ sort.hip
// LSD radix sort of (u64 key, u32 value) pairs on bits [0, nbits).  Result lands in (keys_out, vals_out) — or, when the caller
// passes keys_res / vals_res, in whichever of the two buffer pairs the last pass wrote (an even number of passes ends in the
// "in" pair: no copy; the pointers are returned).  Stable.  Both pairs are clobbered.  Scratch comes from ctx->sort_arena.
// n_dev (nullable): device-resident element count (<= n); n then only sizes the grid and the scratch.
int radix_sort_pairs(Ctx* ctx, uint64_t* keys_in, uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out,
                     int64_t n, int nbits, hipStream_t stream, const int64_t* n_dev = nullptr, uint64_t** keys_res = nullptr,
                     uint32_t** vals_res = nullptr);
size_t radix_sort_scratch_bytes(int64_t n);
// Per-scan variant: every segment [off[b], off[b+1]) (DEVICE int64 offsets, B+1) is sorted on its own on bits [0, nbits), 9 bits
// per pass — no pass is spent on the batch index of contiguous scans.  The result lands in whichever pair the last pass wrote.
int radix_sort_segments(Ctx* ctx, uint64_t* keys_in, uint32_t* vals_in, uint64_t* keys_out, uint32_t* vals_out, int64_t n,
                        const int64_t* off_dev, int B, int nbits, hipStream_t stream, uint64_t** keys_res, uint32_t** vals_res,
                        int idx_bits = 0);
size_t radix_sort_segments_scratch_bytes(int64_t n, int B);
int radix_sort_segments_layout(Ctx* ctx, int64_t n, int B, int32_t** tilehist, int32_t** scanhist);
int radix_sort_segments_passes(int nbits);
int radix_sort_segments_digits();

// ------------------------------------------------------------------ coords.hip
int plan_from_points(Ctx* ctx, const float* points, const int64_t* scan_offsets, int64_t n_cap, int offsets_on_device,
                     int B, int mode, const float* step, hipStream_t stream);
int plan_reserve(Ctx* ctx, int64_t max_points, int B, const int64_t* level_caps);
int plan_sync(Ctx* ctx, hipStream_t stream);      // [SYNC] host copy of the level sizes / error flags of a reserved plan
int plan_from_coords(Ctx* ctx, const int32_t* coords, int64_t n, int B, hipStream_t stream);
int plan_level_coords(Ctx* ctx, int level, int32_t* out, hipStream_t stream);
int count_map_pairs(Ctx* ctx, hipStream_t stream);   // fills dev_pairs[1..7] from the k=3 tables (profiling only)

}  // namespace egonn
