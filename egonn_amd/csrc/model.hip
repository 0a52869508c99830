// C ABI (include/egonn_hip.h) + the EgoNN graph executor.
//
// The graph restates model_factory('egonn') (reference models/model_factory.py:31-76) and
// MinkGL.forward in eval mode (models/minkgl.py:267-315): MinkTrunk (:136-153), ECABasicBlock
// (layers/eca_block.py:56-73), MinkHead (:46-60), DescriptorDecoder / KeypointRegressor / SigmaRegressor
// (:175-225), GeM (layers/pooling.py:82-86).  Weights are addressed by the reference's state_dict keys.
#include <map>
#include <string>

#include "../../include/egonn_hip.h"
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

using namespace egonn;

#define API extern "C" __attribute__((visibility("default")))

struct egonn_ctx : public Ctx {
  // scratch kept between egonn_forward and its readers
  hipStream_t plan_stream = nullptr; // stream the current plan was enqueued on (lazy size queries synchronise it)

  void* level_feat[EGONN_NUM_LEVELS] = {};
  int level_ch[EGONN_NUM_LEVELS] = {};
  int level_bf16 = 0;                // precision of level_feat (last forward)
  bool from_points = false;
};

namespace {

struct TensorRef {
  const float* p = nullptr;
  std::vector<int64_t> shape;
};

struct BnRef {
  const float *w = nullptr, *b = nullptr, *rm = nullptr, *rv = nullptr;
  float *scale = nullptr, *shift = nullptr;
  int c = 0;
};

struct BlockRef {
  const float *conv1 = nullptr, *conv2 = nullptr, *down = nullptr, *eca = nullptr;
  BnRef n1, n2, dn;
  int cin = 0, cout = 0, eca_k = 0;
};

struct MlpRef {
  const float *w0 = nullptr, *b0 = nullptr, *w1 = nullptr, *b1 = nullptr;
  int cin = 0, mid = 0, cout = 0;
};

const int PLANES[7] = {32, 64, 64, 128, 128, 128, 128};   // models/model_factory.py:40
const int GLOBAL_CH = 128, GLOBAL_DIM = 256, LOCAL_CH = 64, LOCAL_DIM = 128;

}  // namespace

struct egonn_model {
  std::map<std::string, TensorRef> t;
  bool ready = false;
  float* folded = nullptr;      // scale/shift storage
  size_t folded_cap = 0;
  // resolved views
  const float* conv0 = nullptr;
  BnRef bn[8];
  const float* convs[8] = {};
  BlockRef blk[8];
  const float *g1x1[8] = {}, *gt[8] = {}, *l1x1[8] = {}, *lt[8] = {};
  const float* gem_p = nullptr;
  MlpRef gdec, ldec, kp, sg;
  // sparse-conv kernels repacked into MFMA fragment order (one buffer, carved in finalize)
  float* packed = nullptr;
  size_t packed_cap = 0;
  void* conv0_unit = nullptr;   // conv0_pack_unit(conv0): 24 KB
  void* lh_pack = nullptr;      // local_heads_pack: the heads' six Linear kernels as fp16 hi | lo fragments (92 KB)
  const float** lh_ptrs = nullptr;   // device array of the six weight pointers (the packer's input)
  const float *p_convs[8] = {}, *p_c1[8] = {}, *p_c2[8] = {}, *p_gt[8] = {}, *p_lt[8] = {};
  // the same kernels packed as bf16 (EGONN_FLAG_BF16): [0] = fp32 set, [1] = bf16 set
  const float *q_convs[8] = {}, *q_c1[8] = {}, *q_c2[8] = {}, *q_gt[8] = {}, *q_lt[8] = {};
  // the same kernels as hi|mid|lo bf16 fragments for the split-bf16 fp32 path (sconv_split.hip)
  const float *s_convs[8] = {}, *s_c1[8] = {}, *s_c2[8] = {}, *s_gt[8] = {}, *s_lt[8] = {};
};

// ------------------------------------------------------------------------------------------ lifecycle
API const char* egonn_last_error(void) { return last_error(); }

API int egonn_debug_set_naive_conv(egonn_ctx* c, int on) {
  EGONN_REQUIRE(c, EGONN_ERR_INVALID, "debug_set_naive_conv: null context");
  if (on >= 1000) { c->conv_variant = on; return EGONN_OK; }     // 1000 + cfg: the split kernel with an explicit configuration
  c->conv_variant = (on == 1) ? 3 : (on == 2 ? 1 : (on == 4 ? 2 : (on == 8 ? 4 : (on == 16 ? 5 : (on == 32 ? 6 : (on == 128 ? 9 : 0))))));
  return EGONN_OK;
}

// Offset-split rule of this context's fp32 sparse convolutions (tests / A-B measurements; sconv_ksplit_rule): map_class 0 = the
// k=3 maps, 1 = the 8-slot maps (k=2,s=2 and transposed); kparts = offset parts as separate workgroups + reducer launch (1 = none),
// kw = offset parts inside a workgroup (0 / 1 = none, 2..4), col_parts = column parts per task (0 = automatic).  -1 keeps a field.
API int egonn_debug_set_ksplit(egonn_ctx* c, int map_class, int level, int kparts, int kw, int col_parts) {
  EGONN_REQUIRE(c && (map_class == 0 || map_class == 1) && level >= 0 && level < EGONN_NUM_LEVELS, EGONN_ERR_INVALID,
                "debug_set_ksplit: bad argument");
  EGONN_REQUIRE(kparts <= 27 && kw <= 4 && col_parts <= 4, EGONN_ERR_INVALID, "debug_set_ksplit: kparts <= 27, kw <= 4, col_parts <= 4");
  if (kparts >= 0) c->ks_rule.kparts[map_class][level] = (int8_t)std::max(kparts, 1);
  if (kw >= 0) c->ks_rule.kw[map_class][level] = (int8_t)kw;
  if (col_parts >= 0) c->ks_rule.col_parts[level] = (int8_t)col_parts;
  return EGONN_OK;
}

API int egonn_ctx_create(egonn_ctx** out, int device, int coord_bits) {
  EGONN_REQUIRE(out != nullptr, EGONN_ERR_INVALID, "ctx_create: null out pointer");
  EGONN_REQUIRE(coord_bits >= 10 && coord_bits <= 16, EGONN_ERR_INVALID, "coord_bits=%d outside [10,16]", coord_bits);
  int ndev = 0;
  HIP_CHECK(hipGetDeviceCount(&ndev));
  EGONN_REQUIRE(device >= 0 && device < ndev, EGONN_ERR_INVALID, "device %d not present (%d HIP devices)", device, ndev);
  HIP_CHECK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, device));
  EGONN_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, EGONN_ERR_INVALID,
                "libegonn_hip is built for gfx950 (MI355X) only; device %d is %s", device, prop.gcnArchName);
  egonn_ctx* c = new egonn_ctx();
  c->device = device;
  c->coord_bits = coord_bits;
  // pinned staging: counts + flags (32), per-level sample offsets, and the scan offsets of egonn_voxelize (int64)
  const size_t hc = sizeof(int32_t) * (32 + (size_t)EGONN_NUM_LEVELS * (EGONN_MAX_BATCH + 1)) + sizeof(int64_t) * (EGONN_MAX_BATCH + 2);
  if (hipHostMalloc(reinterpret_cast<void**>(&c->host_counts), hc) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&c->dev_counts), sizeof(int32_t) * 32) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&c->dev_pairs), sizeof(unsigned long long) * 16) != hipSuccess) {
    set_error("ctx_create: allocation failed");
    delete c;
    return EGONN_ERR_HIP;
  }
  c->dev_flags = c->dev_counts + 16;   // counts[0..11], flags at [16]: fetched by one copy
  sconv_ksplit_defaults(&c->ks_rule);
  if (conv0_lut_init(c) != EGONN_OK) {
    egonn_ctx_destroy(c);
    return EGONN_ERR_HIP;
  }
  *out = c;
  return EGONN_OK;
}

API void egonn_ctx_destroy(egonn_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  c->plan_arena.release();
  c->work_arena.release();
  c->sort_arena.release();
  if (c->host_counts) (void)hipHostFree(c->host_counts);
  if (c->dev_counts) (void)hipFree(c->dev_counts);
  if (c->dev_pairs) (void)hipFree(c->dev_pairs);
  if (c->conv0_lut) (void)hipFree(c->conv0_lut);

  for (auto& r : c->prof.recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  for (auto& r : c->prof.graph_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  for (auto e : c->prof.pool) (void)hipEventDestroy(e);
  delete c;
}

// ------------------------------------------------------------------------------------------ plan
API int egonn_voxelize(egonn_ctx* c, const float* points, const int64_t* scan_offsets, int B, int mode,
                       const float* step, void* stream) {
  EGONN_REQUIRE(c && points && scan_offsets && step, EGONN_ERR_INVALID, "voxelize: null argument");
  HIP_CHECK(hipSetDevice(c->device));
  for (int l = 0; l < EGONN_NUM_LEVELS; ++l) c->level_feat[l] = nullptr;
  c->plan_stream = (hipStream_t)stream;
  EGONN_TRY(plan_from_points(c, points, scan_offsets, 0, 0, B, mode, step, (hipStream_t)stream));
  c->from_points = true;
  return EGONN_OK;
}

API int egonn_ctx_reserve(egonn_ctx* c, int64_t max_points, int batch_size, const int64_t* level_capacity) {
  EGONN_REQUIRE(c, EGONN_ERR_INVALID, "ctx_reserve: null context");
  HIP_CHECK(hipSetDevice(c->device));
  for (int l = 0; l < EGONN_NUM_LEVELS; ++l) c->level_feat[l] = nullptr;
  return plan_reserve(c, max_points, batch_size, level_capacity);
}

API int egonn_voxelize_device(egonn_ctx* c, const float* points, int64_t n_rows, const int64_t* scan_offsets_dev, int B,
                              int mode, const float* step, void* stream) {
  EGONN_REQUIRE(c && points && scan_offsets_dev && step, EGONN_ERR_INVALID, "voxelize_device: null argument");
  HIP_CHECK(hipSetDevice(c->device));
  for (int l = 0; l < EGONN_NUM_LEVELS; ++l) c->level_feat[l] = nullptr;
  c->plan_stream = (hipStream_t)stream;
  EGONN_TRY(plan_from_points(c, points, scan_offsets_dev, n_rows, 1, B, mode, step, (hipStream_t)stream));
  c->from_points = true;
  return EGONN_OK;
}

API int egonn_level_capacity(egonn_ctx* c, int level, int64_t* cap) {
  EGONN_REQUIRE(c && c->plan.valid, EGONN_ERR_STATE, "no coordinate plan (call egonn_voxelize / egonn_coords_set first)");
  EGONN_REQUIRE(level >= 0 && level < EGONN_NUM_LEVELS && cap, EGONN_ERR_INVALID, "level %d out of range", level);
  *cap = c->plan.cap[level];
  return EGONN_OK;
}

// ---- hipGraph capture of a sequence of library calls (thin wrappers, so that a host without a HIP binding can use them)
struct egonn_graph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};
API int egonn_graph_begin(void* stream) {
  HIP_CHECK(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return EGONN_OK;
}
API int egonn_graph_end(void* stream, egonn_graph** out) {
  EGONN_REQUIRE(out, EGONN_ERR_INVALID, "graph_end: null out pointer");
  egonn_graph* g = new egonn_graph();
  hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g->graph);
  if (e == hipSuccess) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    set_error("graph_end: %s (a captured call allocated or synchronised: reserve the context and run the sequence once "
              "before capturing)", hipGetErrorString(e));
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return EGONN_ERR_HIP;
  }
  *out = g;
  return EGONN_OK;
}
API int egonn_graph_launch(egonn_graph* g, void* stream) {
  EGONN_REQUIRE(g && g->exec, EGONN_ERR_INVALID, "graph_launch: null graph");
  HIP_CHECK(hipGraphLaunch(g->exec, (hipStream_t)stream));
  return EGONN_OK;
}
API void egonn_graph_destroy(egonn_graph* g) {
  if (!g) return;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
}

API int egonn_plan_status(egonn_ctx* c, void* stream) {
  EGONN_REQUIRE(c && (c->plan.valid || c->plan.built_reserved), EGONN_ERR_STATE,
                "no coordinate plan (call egonn_voxelize / egonn_coords_set first)");
  HIP_CHECK(hipSetDevice(c->device));
  if (c->plan.built_reserved) c->plan.valid = true;   // a replayed graph rebuilt the plan behind the host's back
  c->plan.exact = false;                 // read the state again: sizes (reserved plans) and the flags launches raise after the
  return plan_sync(c, (hipStream_t)stream);   // plan was built (range guard of the fp16-split convolutions)
}

// fp32 feature maps: on = 1 runs every sparse convolution of this context on the exact fp32 kernels (v_mfma_f32_16x16x4_f32, the
// full fp32 range), on = 0 (default) the levels <= 5 on the fp16-split matrix pipe (sconv_split.hip: |activation| < 65504, guarded:
// egonn_plan_status reports EGONN_STATUS_FP16_RANGE).  The choice is per context and a function of the layer, never of the batch.
// on = 1: every fp32 sparse convolution of this context on the fp16-split pipe first takes max |input| (one reduction launch) and scales
// the gathered rows by the power of two that puts it into [2^13, 2^14), undone exactly in the epilogue — what the kernels do to their
// weights.  For operands far below 1 (the input-gradient convolutions of a training step, egonn_amd/train.py): an fp16 low part
// flushes below 2^-25 and carries 2^-25 absolute error below 2^-14.  Eager plans only (the element count comes from the host).
API int egonn_ctx_set_operand_autoscale(egonn_ctx* c, int on) {
  EGONN_REQUIRE(c && (on == 0 || on == 1), EGONN_ERR_INVALID, "ctx_set_operand_autoscale: bad argument");
  c->operand_autoscale = on;
  return EGONN_OK;
}

// on = 1: egonn_forward materialises the block output of EVERY level (egonn_forward_level_features(ctx, 1, ...) then has a map to
// return): level 1's block tail runs as its own launch instead of inside level 2's strided convolution (bitwise the same result).
API int egonn_debug_keep_level_features(egonn_ctx* c, int on) {
  EGONN_REQUIRE(c && (on == 0 || on == 1), EGONN_ERR_INVALID, "debug_keep_level_features: bad argument");
  c->keep_level_features = on;
  return EGONN_OK;
}

API int egonn_ctx_set_exact_fp32(egonn_ctx* c, int on) {
  EGONN_REQUIRE(c && (on == 0 || on == 1), EGONN_ERR_INVALID, "ctx_set_exact_fp32: bad argument");
  c->split_max_level = on ? -1 : 5;
  return EGONN_OK;
}

API int egonn_coords_set(egonn_ctx* c, const int32_t* coords, int64_t n, int B, void* stream) {
  EGONN_REQUIRE(c && coords, EGONN_ERR_INVALID, "coords_set: null argument");
  HIP_CHECK(hipSetDevice(c->device));
  for (int l = 0; l < EGONN_NUM_LEVELS; ++l) c->level_feat[l] = nullptr;
  c->plan_stream = (hipStream_t)stream;
  EGONN_TRY(plan_from_coords(c, coords, n, B, (hipStream_t)stream));
  c->from_points = false;
  return EGONN_OK;
}

#define REQUIRE_PLAN(c)                                                                               \
  EGONN_REQUIRE((c) && (c)->plan.valid, EGONN_ERR_STATE, "no coordinate plan (call egonn_voxelize / " \
                                                         "egonn_coords_set first)")

API int egonn_level_count(egonn_ctx* c, int level, int64_t* n) {
  REQUIRE_PLAN(c);
  EGONN_REQUIRE(level >= 0 && level < EGONN_MAX_LEVELS && n, EGONN_ERR_INVALID, "level %d out of range", level);
  EGONN_TRY(plan_sync(c, c->plan_stream));
  *n = c->plan.lv[level].n;
  return EGONN_OK;
}

API int egonn_level_batch_offsets(egonn_ctx* c, int level, int64_t* off) {
  REQUIRE_PLAN(c);
  EGONN_REQUIRE(level >= 0 && level < EGONN_NUM_LEVELS && off, EGONN_ERR_INVALID, "level %d out of range", level);
  EGONN_TRY(plan_sync(c, c->plan_stream));
  for (int b = 0; b <= c->plan.batch; ++b) off[b] = c->plan.boff_host[level][b];
  return EGONN_OK;
}

API int egonn_level_coords(egonn_ctx* c, int level, int32_t* out, void* stream) {
  REQUIRE_PLAN(c);
  HIP_CHECK(hipSetDevice(c->device));
  return plan_level_coords(c, level, out, (hipStream_t)stream);
}

__global__ void input_index_kernel(const int32_t* __restrict__ perm0, const uint64_t* __restrict__ keys, int64_t n,
                                   int bshift, const int64_t* __restrict__ scan_off, int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t v = perm0[i];
  if (scan_off) v -= scan_off[(int)(keys[i] >> bshift)];
  out[i] = v;
}

API int egonn_input_index(egonn_ctx* c, int64_t* out, void* stream) {
  REQUIRE_PLAN(c);
  HIP_CHECK(hipSetDevice(c->device));
  EGONN_TRY(plan_sync(c, c->plan_stream));
  const Level& L = c->plan.lv[0];
  if (L.n == 0) return EGONN_OK;
  hipLaunchKernelGGL(input_index_kernel, dim3((unsigned)cdiv(L.n, 256)), dim3(256), 0, (hipStream_t)stream,
                     c->plan.perm0, L.keys, L.n, 3 * c->plan.coord_bits, c->from_points ? c->plan.scan_off : nullptr,
                     out);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// ------------------------------------------------------------------------------------------ operators
// scratch for the stand-alone operator entry points (the forward carves its own from the work arena)
static float* op_scratch(egonn_ctx* c) {
  for (int l = 0; l < EGONN_NUM_LEVELS; ++l) c->level_feat[l] = nullptr;
  const size_t ks = sconv_ksplit_scratch_floats(c);
  c->ks_part = nullptr;
  c->ks_part_floats = 0;
  if (c->work_arena.ensure((SCONV_SCRATCH_FLOATS + ks) * sizeof(float) + 8192) != EGONN_OK) return nullptr;
  c->work_arena.reset();
  float* scratch = c->work_arena.alloc<float>(SCONV_SCRATCH_FLOATS);
  if (ks) {
    c->ks_part = c->work_arena.alloc<float>(ks);
    c->ks_part_floats = c->ks_part ? ks : 0;
  }
  return scratch;
}

API int egonn_conv(egonn_ctx* c, int level_in, int level_out, int ks, const float* in, int cin, const float* kernel,
                   int cout, const float* scale, const float* shift, int relu, float* out, void* stream) {
  REQUIRE_PLAN(c);
  HIP_CHECK(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  EGONN_REQUIRE(level_in >= 0 && level_in < EGONN_NUM_LEVELS && level_out >= level_in && level_out < EGONN_NUM_LEVELS,
                EGONN_ERR_INVALID, "conv: levels (%d,%d) out of range", level_in, level_out);
  const Plan& P = c->plan;
  if (ks == 1) {
    EGONN_REQUIRE(level_in == level_out, EGONN_ERR_INVALID, "1x1 conv cannot change the level");
    return dense_forward(in, P.lv[level_in].n, cin, kernel, 0, cout, nullptr, scale, shift, relu ? ACT_RELU : ACT_NONE,
                         nullptr, out, st);
  }
  if (ks == 5) {
    EGONN_REQUIRE(level_in == 0 && level_out == 0 && cin == 1, EGONN_ERR_INVALID,
                  "k=5 convolution is implemented for the stride-1 input layer with Cin=1 only");
    return conv0_k5_forward(c, in, kernel, cout, scale, shift, relu, out, 0, st);
  }
  if (ks == 3) {
    EGONN_REQUIRE(level_in == level_out && level_in >= 1, EGONN_ERR_INVALID,
                  "k=3 convolution is implemented for levels 1..7 (same in/out level)");
    return sconv_map(c, 0, level_out, in, kernel, nullptr, nullptr, cin, cout, 0, scale, shift, relu, out, nullptr, op_scratch(c),
                     SCONV_SCRATCH_FLOATS, st);
  }
  if (ks == 2) {
    EGONN_REQUIRE(level_out == level_in + 1, EGONN_ERR_INVALID, "k=2,s=2 convolution maps level l to l+1");
    return sconv_map(c, 1, level_out, in, kernel, nullptr, nullptr, cin, cout, 0, scale, shift, relu, out, nullptr, op_scratch(c),
                     SCONV_SCRATCH_FLOATS, st);
  }
  set_error("conv: kernel_size %d not supported (1, 2, 3, 5)", ks);
  return EGONN_ERR_INVALID;
}

API int egonn_conv_transpose(egonn_ctx* c, int level_in, const float* in, int cin, const float* kernel, int cout,
                             float* out, void* stream) {
  REQUIRE_PLAN(c);
  HIP_CHECK(hipSetDevice(c->device));
  EGONN_REQUIRE(level_in >= 1 && level_in < EGONN_NUM_LEVELS, EGONN_ERR_INVALID,
                "transposed conv: input level %d out of range [1,7]", level_in);
  if (level_in == 1) EGONN_TRY(ensure_level0_parent_table(c, (hipStream_t)stream));
  return sconv_map(c, 2, level_in - 1, in, kernel, nullptr, nullptr, cin, cout, 0, nullptr, nullptr, 0, out, nullptr, op_scratch(c),
                   SCONV_SCRATCH_FLOATS, (hipStream_t)stream);
}

// Sparse convolution on a map of the plan with explicit precision (the operator behind egonn_conv / egonn_conv_transpose).
API int egonn_sparse_conv(egonn_ctx* c, int map_kind, int level_out, const void* in, int cin, const float* kernel, int cout,
                          int bf16, const float* scale, const float* shift, int relu, void* out, float* group_sums,
                          void* stream) {
  REQUIRE_PLAN(c);
  HIP_CHECK(hipSetDevice(c->device));
  EGONN_REQUIRE(in && kernel && out, EGONN_ERR_INVALID, "sparse_conv: null argument");
  return sconv_map(c, map_kind, level_out, in, kernel, nullptr, nullptr, cin, cout, bf16, scale, shift, relu, out, group_sums,
                   op_scratch(c), SCONV_SCRATCH_FLOATS, (hipStream_t)stream);
}
// Row-group tables of every kernel map of the plan a training step (or a sequence of stand-alone operator calls) uses, in ONE
// launch: k=3 and k=2,s=2 maps of levels 1..7, transposed maps onto levels 0..6 (with_level0_transpose: the input gradient of the
// first strided convolution; builds the level-0 parent table).  Without it every operator builds its map's tables on first
// use — 21 launches of 12 us per training step.
API int egonn_prepare_maps(egonn_ctx* c, int with_level0_transpose, void* stream) {
  REQUIRE_PLAN(c);
  HIP_CHECK(hipSetDevice(c->device));
  int kinds[RG_MAX_JOBS], levels[RG_MAX_JOBS], n = 0;
  for (int l = 1; l < EGONN_NUM_LEVELS; ++l) { kinds[n] = 0; levels[n++] = l; }
  for (int l = 1; l < EGONN_NUM_LEVELS; ++l) { kinds[n] = 1; levels[n++] = l; }
  for (int l = with_level0_transpose ? 0 : 1; l < EGONN_NUM_LEVELS - 1; ++l) { kinds[n] = 2; levels[n++] = l; }
  return ensure_rowgroups(c, kinds, levels, n, (hipStream_t)stream);
}

// Measurement hook (tools/sconv_trace.py): device buffer the traced conv build (debug variant 128) writes its per-task
// timestamps into (8 u64 per wave task); null switches it off.
API int egonn_debug_set_trace(void* buf) {
  g_sconv_trace = reinterpret_cast<unsigned long long*>(buf);
  return EGONN_OK;
}

// Measurement hook (tools/rowgroup_stats.py): device copies of a map's row-group tables.  gmask_out: [groups] u32;
// snbr_out: [groups][K][16] i32 (nullable).  Returns the number of groups copied (<= capacity_groups).  [SYNC]
API int egonn_debug_rowgroup_tables(egonn_ctx* c, int map_kind, int level_out, uint32_t* gmask_out, int32_t* snbr_out,
                                    int64_t capacity_groups, int64_t* n_groups, void* stream) {
  REQUIRE_PLAN(c);
  HIP_CHECK(hipSetDevice(c->device));
  EGONN_REQUIRE(n_groups && gmask_out, EGONN_ERR_INVALID, "rowgroup_tables: null argument");
  EGONN_TRY(ensure_rowgroups(c, &map_kind, &level_out, 1, (hipStream_t)stream));
  const Level& V = c->plan.lv[level_out];
  const RowGroups& rg = map_kind == 0 ? V.rg27 : (map_kind == 1 ? V.rg8 : V.rgT);
  int32_t ng = 0;
  HIP_CHECK(hipMemcpyAsync(&ng, rg.meta, sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  EGONN_REQUIRE(ng <= capacity_groups, EGONN_ERR_INVALID, "rowgroup_tables: %d groups, room for %lld", ng, (long long)capacity_groups);
  HIP_CHECK(hipMemcpyAsync(gmask_out, rg.gmask, (size_t)ng * sizeof(uint32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (snbr_out)
    HIP_CHECK(hipMemcpyAsync(snbr_out, rg.snbr, (size_t)ng * rg.K * 16 * sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  *n_groups = ng;
  return EGONN_OK;
}

// Number of row groups (16 output rows each) of a map: rows of the `group_sums` output of egonn_sparse_conv; the groups of
// sample b are [first_group[b], first_group[b+1]) (HOST copy, B+1 entries).  [SYNC]
API int egonn_map_groups(egonn_ctx* c, int map_kind, int level_out, int64_t* n_groups, int64_t* first_group, void* stream) {
  REQUIRE_PLAN(c);
  HIP_CHECK(hipSetDevice(c->device));
  EGONN_REQUIRE(n_groups, EGONN_ERR_INVALID, "map_groups: null argument");
  EGONN_TRY(ensure_rowgroups(c, &map_kind, &level_out, 1, (hipStream_t)stream));
  const Level& V = c->plan.lv[level_out];
  const RowGroups& rg = map_kind == 0 ? V.rg27 : (map_kind == 1 ? V.rg8 : V.rgT);
  std::vector<int32_t> h((size_t)c->plan.batch + 2);
  HIP_CHECK(hipMemcpyAsync(h.data(), rg.meta, h.size() * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  *n_groups = h[0];
  if (first_group)
    for (int b = 0; b <= c->plan.batch; ++b) first_group[b] = h[1 + b];
  return EGONN_OK;
}

__global__ void avg_finish_kernel(const float* __restrict__ partial, const int32_t* __restrict__ boff, int c,
                                  float* __restrict__ out) {
  const int b = blockIdx.x, t = threadIdx.x;
  if (t >= c) return;
  const int32_t n = boff[b + 1] - boff[b];
  float s = 0.f;
  for (int ch = 0; ch < SEG_CHUNKS; ++ch) s += partial[((int64_t)b * SEG_CHUNKS + ch) * c + t];
  out[(int64_t)b * c + t] = n > 0 ? s / (float)n : 0.f;
}

API int egonn_global_avg_pool(egonn_ctx* c, int level, const float* in, int ch, float* out, void* stream) {
  REQUIRE_PLAN(c);
  HIP_CHECK(hipSetDevice(c->device));
  EGONN_REQUIRE(level >= 0 && level < EGONN_NUM_LEVELS, EGONN_ERR_INVALID, "level %d out of range", level);
  const int B = c->plan.batch;
  for (int l = 0; l < EGONN_NUM_LEVELS; ++l) c->level_feat[l] = nullptr;
  EGONN_TRY(c->work_arena.ensure((size_t)B * SEG_CHUNKS * ch * 4 + 4096));
  c->work_arena.reset();
  float* partial = c->work_arena.alloc<float>((size_t)B * SEG_CHUNKS * ch);
  EGONN_TRY(segment_partial_sums(in, c->plan.lv[level].boff, B, ch, 0, nullptr, partial, (hipStream_t)stream));
  hipLaunchKernelGGL(avg_finish_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, partial, c->plan.lv[level].boff, ch,
                     out);
  HIP_CHECK(hipGetLastError());
  return EGONN_OK;
}

// eval-mode MinkowskiBatchNorm folded to scale/shift (nn.BatchNorm1d eps as given): scale = w / sqrt(var + eps)
API int egonn_bn_fold(const float* weight, const float* bias, const float* running_mean, const float* running_var,
                      float eps, int channels, float* scale, float* shift, void* stream) {
  EGONN_REQUIRE(weight && bias && running_mean && running_var && scale && shift && channels > 0, EGONN_ERR_INVALID,
                "bn_fold: bad argument");
  return bn_fold(weight, bias, running_mean, running_var, eps, channels, scale, shift, (hipStream_t)stream);
}

// Tail of a residual block on level `level`:  out = relu(x * gate + residual)
//   eca_weight != NULL: gate = sigmoid(conv1d_k(per-sample mean of x))  (ECABasicBlock, layers/eca_block.py:21-36,66-71)
//   eca_weight == NULL: gate = 1                                        (ME BasicBlock: out += residual; relu)
API int egonn_block_tail(egonn_ctx* c, int level, const float* x, const float* residual, int channels,
                         const float* eca_weight, int eca_ksize, float* out, void* stream) {
  REQUIRE_PLAN(c);
  EGONN_REQUIRE(level >= 0 && level < EGONN_NUM_LEVELS && x && residual && out, EGONN_ERR_INVALID, "block_tail: bad argument");
  HIP_CHECK(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  const int B = c->plan.batch;
  const Level& L = c->plan.lv[level];
  for (int l = 0; l < EGONN_NUM_LEVELS; ++l) c->level_feat[l] = nullptr;
  EGONN_TRY(c->work_arena.ensure(((size_t)B * SEG_CHUNKS * channels + (size_t)B * channels + 64) * 4 + 4096));
  c->work_arena.reset();
  float* partial = c->work_arena.alloc<float>((size_t)B * SEG_CHUNKS * channels + (size_t)B * channels);
  EGONN_REQUIRE(partial, EGONN_ERR_STATE, "work arena too small");
  float ones[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  (void)ones;
  if (eca_weight) {
    EGONN_TRY(segment_partial_sums(x, L.boff, B, channels, 0, nullptr, partial, st));
    return eca_apply(x, residual, partial, L.boff, B, L.n, channels, eca_weight, eca_ksize, out, st);
  }
  return add_act(x, residual, L.n * channels, 1, out, st);
}

// SparseTensor + SparseTensor on the same coordinate map (models/minkfpn.py:91, minkgl.py:56): out = a + b
API int egonn_add(const float* a, const float* b, int64_t n, float* out, void* stream) {
  EGONN_REQUIRE(a && b && out && n >= 0, EGONN_ERR_INVALID, "add: bad argument");
  return add_act(a, b, n, 0, out, (hipStream_t)stream);
}

// Features handed in in the caller's row order -> plan (Z-order) row order: out[i] = features[input_index[i]]
API int egonn_gather_input(egonn_ctx* c, const float* features, int channels, float* out, void* stream) {
  REQUIRE_PLAN(c);
  EGONN_REQUIRE(features && out && channels >= 1, EGONN_ERR_INVALID, "gather_input: bad argument");
  HIP_CHECK(hipSetDevice(c->device));
  EGONN_REQUIRE(!c->from_points, EGONN_ERR_STATE, "gather_input: voxelize plans have no caller row order");
  return gather_rows(features, c->plan.perm0, c->plan.lv[0].n, channels, out, (hipStream_t)stream);
}

// GeM pooling over the rows of level `level` (layers/pooling.py:82-86): out (B, channels)
API int egonn_gem(egonn_ctx* c, int level, const float* x, int channels, const float* p, float* out, void* stream) {
  REQUIRE_PLAN(c);
  EGONN_REQUIRE(level >= 0 && level < EGONN_NUM_LEVELS && x && p && out, EGONN_ERR_INVALID, "gem: bad argument");
  HIP_CHECK(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  const int B = c->plan.batch;
  for (int l = 0; l < EGONN_NUM_LEVELS; ++l) c->level_feat[l] = nullptr;
  EGONN_TRY(c->work_arena.ensure((size_t)B * SEG_CHUNKS * channels * 4 + 4096));
  c->work_arena.reset();
  float* partial = c->work_arena.alloc<float>((size_t)B * SEG_CHUNKS * channels);
  EGONN_REQUIRE(partial, EGONN_ERR_STATE, "work arena too small");
  EGONN_TRY(segment_partial_sums(x, c->plan.lv[level].boff, B, channels, 1, p, partial, st));
  return gem_finish(partial, c->plan.lv[level].boff, B, channels, p, out, st);
}

// ------------------------------------------------------------------------------------------ model
API int egonn_model_create(egonn_model** m) {
  EGONN_REQUIRE(m, EGONN_ERR_INVALID, "model_create: null out pointer");
  *m = new egonn_model();
  return EGONN_OK;
}

API void egonn_model_destroy(egonn_model* m) {
  if (!m) return;
  if (m->folded) (void)hipFree(m->folded);
  if (m->packed) (void)hipFree(m->packed);
  if (m->conv0_unit) (void)hipFree(m->conv0_unit);
  if (m->lh_pack) (void)hipFree(m->lh_pack);
  if (m->lh_ptrs) (void)hipFree(m->lh_ptrs);
  delete m;
}

API int egonn_model_set_tensor(egonn_model* m, const char* key, const float* data, int ndim, const int64_t* shape) {
  EGONN_REQUIRE(m && key && data && ndim >= 0 && ndim <= 4, EGONN_ERR_INVALID, "model_set_tensor: bad argument");
  TensorRef r;
  r.p = data;
  r.shape.assign(shape, shape + ndim);
  m->t[key] = r;
  m->ready = false;
  return EGONN_OK;
}

namespace {

int get_tensor(egonn_model* m, const std::string& key, std::initializer_list<int64_t> want, const float** out) {
  auto it = m->t.find(key);
  EGONN_REQUIRE(it != m->t.end(), EGONN_ERR_STATE, "model: missing state_dict tensor '%s'", key.c_str());
  const std::vector<int64_t>& s = it->second.shape;
  bool ok = s.size() == want.size();
  size_t i = 0;
  for (int64_t w : want) {
    if (ok && s[i] != w) ok = false;
    ++i;
  }
  if (!ok) {
    std::string got = "(", exp = "(";
    for (int64_t v : s) got += std::to_string(v) + ",";
    for (int64_t v : want) exp += std::to_string(v) + ",";
    set_error("model: tensor '%s' has shape %s), expected %s)", key.c_str(), got.c_str(), exp.c_str());
    return EGONN_ERR_INVALID;
  }
  *out = it->second.p;
  return EGONN_OK;
}

int get_bn(egonn_model* m, const std::string& prefix, int c, BnRef* bn, float** cursor) {
  bn->c = c;
  EGONN_TRY(get_tensor(m, prefix + ".bn.weight", {c}, &bn->w));
  EGONN_TRY(get_tensor(m, prefix + ".bn.bias", {c}, &bn->b));
  EGONN_TRY(get_tensor(m, prefix + ".bn.running_mean", {c}, &bn->rm));
  EGONN_TRY(get_tensor(m, prefix + ".bn.running_var", {c}, &bn->rv));
  bn->scale = *cursor;
  bn->shift = *cursor + c;
  *cursor += 2 * c;
  return EGONN_OK;
}

int get_mlp(egonn_model* m, const std::string& prefix, int cin, int mid, int cout, MlpRef* r) {
  r->cin = cin; r->mid = mid; r->cout = cout;
  EGONN_TRY(get_tensor(m, prefix + ".net.0.linear.weight", {mid, cin}, &r->w0));
  EGONN_TRY(get_tensor(m, prefix + ".net.0.linear.bias", {mid}, &r->b0));
  EGONN_TRY(get_tensor(m, prefix + ".net.2.linear.weight", {cout, mid}, &r->w1));
  EGONN_TRY(get_tensor(m, prefix + ".net.2.linear.bias", {cout}, &r->b1));
  return EGONN_OK;
}

int fold(const BnRef& bn, hipStream_t st) { return bn_fold(bn.w, bn.b, bn.rm, bn.rv, 1e-5f, bn.c, bn.scale, bn.shift, st); }

}  // namespace

API int egonn_model_finalize(egonn_model* m, void* stream) {
  EGONN_REQUIRE(m, EGONN_ERR_INVALID, "model_finalize: null model");
  hipStream_t st = (hipStream_t)stream;
  const size_t need = 2 * 128 * 32;   // generous: 24 BN layers x <=128 ch x (scale, shift)
  if (m->folded_cap < need) {
    if (m->folded) HIP_CHECK(hipFree(m->folded));
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&m->folded), need * sizeof(float)));
    m->folded_cap = need;
  }
  float* cur = m->folded;
  EGONN_TRY(get_tensor(m, "trunk.convs.0.kernel", {125, 1, 32}, &m->conv0));
  EGONN_TRY(get_bn(m, "trunk.bn.0", 32, &m->bn[0], &cur));
  int inpl = PLANES[0];
  for (int i = 1; i <= 7; ++i) {
    const std::string si = std::to_string(i);
    const int cout = PLANES[i - 1];
    EGONN_TRY(get_tensor(m, "trunk.convs." + si + ".kernel", {8, inpl, inpl}, &m->convs[i]));
    EGONN_TRY(get_bn(m, "trunk.bn." + si, inpl, &m->bn[i], &cur));
    BlockRef& b = m->blk[i];
    b.cin = inpl;
    b.cout = cout;
    const std::string pre = "trunk.blocks." + si + ".0";
    EGONN_TRY(get_tensor(m, pre + ".conv1.kernel", {27, inpl, cout}, &b.conv1));
    EGONN_TRY(get_bn(m, pre + ".norm1", cout, &b.n1, &cur));
    EGONN_TRY(get_tensor(m, pre + ".conv2.kernel", {27, cout, cout}, &b.conv2));
    EGONN_TRY(get_bn(m, pre + ".norm2", cout, &b.n2, &cur));
    if (inpl != cout) {
      EGONN_TRY(get_tensor(m, pre + ".downsample.0.kernel", {inpl, cout}, &b.down));
      EGONN_TRY(get_bn(m, pre + ".downsample.1", cout, &b.dn, &cur));
    } else {
      b.down = nullptr;
    }
    // ECALayer kernel size (layers/eca_block.py:14-15): t = int(|log2(C)+1| / 2), odd
    int tt = 0;
    {
      int lg = 0;
      while ((1 << lg) < cout) ++lg;
      tt = (lg + 1) / 2;
    }
    b.eca_k = (tt % 2) ? tt : tt + 1;
    EGONN_TRY(get_tensor(m, pre + ".eca.conv.weight", {1, 1, b.eca_k}, &b.eca));
    inpl = cout;
  }
  for (int l : {5, 6, 7})
    EGONN_TRY(get_tensor(m, "global_head.conv1x1." + std::to_string(l) + ".kernel", {PLANES[l - 1], GLOBAL_CH}, &m->g1x1[l]));
  for (int l : {6, 7})
    EGONN_TRY(get_tensor(m, "global_head.tconv." + std::to_string(l) + ".kernel", {8, GLOBAL_CH, GLOBAL_CH}, &m->gt[l]));
  for (int l : {3, 4})
    EGONN_TRY(get_tensor(m, "local_head.conv1x1." + std::to_string(l) + ".kernel", {PLANES[l - 1], LOCAL_CH}, &m->l1x1[l]));
  EGONN_TRY(get_tensor(m, "local_head.tconv.4.kernel", {8, LOCAL_CH, LOCAL_CH}, &m->lt[4]));
  m->gem_p = nullptr;                  // GeM exponent; MAC / SPoC pooling (layers/pooling.py:46-69) has no parameter
  if (m->t.count("global_pooling.pooling.p")) EGONN_TRY(get_tensor(m, "global_pooling.pooling.p", {1}, &m->gem_p));
  EGONN_TRY(get_mlp(m, "global_descriptor_decoder", GLOBAL_CH, GLOBAL_DIM + (GLOBAL_CH - GLOBAL_DIM) / 2, GLOBAL_DIM, &m->gdec));
  EGONN_TRY(get_mlp(m, "local_descriptor_decoder", LOCAL_CH, LOCAL_DIM + (LOCAL_CH - LOCAL_DIM) / 2, LOCAL_DIM, &m->ldec));
  EGONN_TRY(get_mlp(m, "local_keypoint_regressor", LOCAL_CH, LOCAL_CH / 2, 3, &m->kp));
  EGONN_TRY(get_mlp(m, "local_sigma_regressor", LOCAL_CH, LOCAL_CH / 2, 1, &m->sg));

  // ---- repack every sparse-conv kernel into item-major MFMA fragment order (sconv.hip), fp32 and bf16
  {
    size_t need_p = 0;
    for (int i = 1; i <= 7; ++i) {
      const BlockRef& b = m->blk[i];
      need_p += (size_t)8 * b.cin * b.cin + (size_t)27 * b.cin * b.cout + (size_t)27 * b.cout * b.cout;
    }
    need_p += (size_t)2 * 8 * GLOBAL_CH * GLOBAL_CH + (size_t)8 * LOCAL_CH * LOCAL_CH;
    if (m->packed_cap < need_p) {
      if (m->packed) HIP_CHECK(hipFree(m->packed));
      HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&m->packed), (3 * need_p + 512) * sizeof(float)));
      m->packed_cap = need_p;
    }
    float* pc = m->packed;
    uint16_t* qc = reinterpret_cast<uint16_t*>(m->packed + need_p);      // bf16 copies behind the fp32 ones
    uint16_t* sc = reinterpret_cast<uint16_t*>(m->packed + need_p + need_p / 2 + 16);   // split fragments behind the bf16 ones
    auto pack2 = [&](const float* w, int K, int ci, int co, const float** dst32, const float** dst16, const float** dsts) -> int {
      EGONN_TRY(pack_rg_weights(w, K, ci, co, 0, 0, 0, pc, st));
      EGONN_TRY(pack_rg_weights(w, K, ci, co, 1, 0, 0, qc, st));
      *dst32 = pc;
      *dst16 = reinterpret_cast<const float*>(qc);
      *dsts = nullptr;
      if (sconv_split_supported(ci, co)) {
        EGONN_TRY(pack_split_weights(w, K, ci, co, 0, 0, sc, st));
        *dsts = reinterpret_cast<const float*>(sc);
        sc += split_weights_bytes(K, ci, co) / 2;
      }
      pc += (size_t)K * ci * co;
      qc += (size_t)K * ci * co;
      return EGONN_OK;
    };
    for (int i = 1; i <= 7; ++i) {
      const BlockRef& b = m->blk[i];
      EGONN_TRY(pack2(m->convs[i], 8, b.cin, b.cin, &m->p_convs[i], &m->q_convs[i], &m->s_convs[i]));
      EGONN_TRY(pack2(b.conv1, 27, b.cin, b.cout, &m->p_c1[i], &m->q_c1[i], &m->s_c1[i]));
      EGONN_TRY(pack2(b.conv2, 27, b.cout, b.cout, &m->p_c2[i], &m->q_c2[i], &m->s_c2[i]));
    }
    EGONN_TRY(pack2(m->gt[6], 8, GLOBAL_CH, GLOBAL_CH, &m->p_gt[6], &m->q_gt[6], &m->s_gt[6]));
    EGONN_TRY(pack2(m->gt[7], 8, GLOBAL_CH, GLOBAL_CH, &m->p_gt[7], &m->q_gt[7], &m->s_gt[7]));
    EGONN_TRY(pack2(m->lt[4], 8, LOCAL_CH, LOCAL_CH, &m->p_lt[4], &m->q_lt[4], &m->s_lt[4]));
  }
  if (!m->conv0_unit) HIP_CHECK(hipMalloc(&m->conv0_unit, 2 * 4 * 3 * 64 * 16));
  EGONN_TRY(conv0_pack_unit(m->conv0, m->conv0_unit, st));
  if (m->ldec.cin == 64 && m->ldec.mid == 96 && m->ldec.cout == 128 && m->kp.mid == 32 && m->sg.mid == 32 && m->kp.cin == 64 &&
      m->sg.cin == 64) {       // the local heads of models/minkgl.py:175-225 in their shipped sizes: split-fp16 fragments
    if (!m->lh_pack) HIP_CHECK(hipMalloc(&m->lh_pack, local_heads_pack_bytes()));
    if (!m->lh_ptrs) HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&m->lh_ptrs), 6 * sizeof(float*)));
    const float* hp[6] = {m->ldec.w0, m->ldec.w1, m->kp.w0, m->kp.w1, m->sg.w0, m->sg.w1};
    HIP_CHECK(hipMemcpyAsync(m->lh_ptrs, hp, sizeof(hp), hipMemcpyHostToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));                    // (hp lives on this stack frame)
    EGONN_TRY(local_heads_pack(m->lh_ptrs, m->lh_pack, st));
  }
  EGONN_TRY(fold(m->bn[0], st));
  for (int i = 1; i <= 7; ++i) {
    EGONN_TRY(fold(m->bn[i], st));
    EGONN_TRY(fold(m->blk[i].n1, st));
    EGONN_TRY(fold(m->blk[i].n2, st));
    if (m->blk[i].down) EGONN_TRY(fold(m->blk[i].dn, st));
  }
  m->ready = true;
  return EGONN_OK;
}

// ------------------------------------------------------------------------------------------ forward
namespace {

// EGONN_DEBUG_SYNC=1: synchronise after every stage of egonn_forward and print its name (a GPU fault aborts the process at
// the next synchronisation: the last name printed is the stage that faulted).  Debug aid only; never set in captures.
bool debug_sync_on() {
  static const bool on = [] { const char* e = getenv("EGONN_DEBUG_SYNC"); return e && e[0] == '1'; }();
  return on;
}
#define DBG_SYNC(...)                                                 \
  do {                                                                \
    if (debug_sync_on()) {                                            \
      fprintf(stderr, "[egonn] " __VA_ARGS__);                        \
      fprintf(stderr, "\n");                                          \
      fflush(stderr);                                                 \
      HIP_CHECK(hipStreamSynchronize(st));                            \
    }                                                                 \
  } while (0)

int run_mlp(const MlpRef& r, const float* x, int64_t n, int act_out, float* hidden, float* out, hipStream_t st,
            const int32_t* n_dev) {
  EGONN_TRY(dense_forward_ex(x, 0, n, r.cin, r.w0, 1, r.mid, r.b0, nullptr, nullptr, ACT_RELU, nullptr, 0, hidden, 0, st, n_dev));
  return dense_forward_ex(hidden, 0, n, r.mid, r.w1, 1, r.cout, r.b1, nullptr, nullptr, act_out, nullptr, 0, out, 0, st, n_dev);
}

}  // namespace

API int egonn_forward(egonn_ctx* c, egonn_model* m, const float* features, int quant_mode, const float* step, int flags,
                      float* out_global, float* out_desc, float* out_kp, float* out_sigma, void* stream) {
  REQUIRE_PLAN(c);
  EGONN_REQUIRE(m && m->ready, EGONN_ERR_STATE, "model not finalized (call egonn_model_finalize)");
  EGONN_REQUIRE(step, EGONN_ERR_INVALID, "forward: null argument");
  HIP_CHECK(hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  Plan& P = c->plan;
  const int B = P.batch;
  const int32_t* cnt = c->dev_counts;                          // device row counts per level (the kernels clip to them)
  const int bf16 = (flags & EGONN_FLAG_BF16) ? 1 : 0;          // feature maps + sparse-conv weights in bf16 (configs[2])
  const size_t es = bf16 ? 2 : 4;                              // bytes per feature-map element
  const bool do_global = !(flags & EGONN_FLAG_DISABLE_GLOBAL);
  const bool do_local = !(flags & EGONN_FLAG_DISABLE_LOCAL);
  EGONN_REQUIRE(!do_global || out_global, EGONN_ERR_INVALID, "forward: out_global is null");
  EGONN_REQUIRE(!do_local || (out_desc && out_kp && out_sigma), EGONN_ERR_INVALID, "forward: local outputs are null");

  // ---- row-group tables of every map the graph uses: one launch per plan
  {
    int kinds[RG_MAX_JOBS], levels[RG_MAX_JOBS], nreq = 0;
    for (int l = 1; l <= 7; ++l) { kinds[nreq] = 0; levels[nreq++] = l; }
    for (int l = 1; l <= 7; ++l) { kinds[nreq] = 1; levels[nreq++] = l; }
    if (do_global) { kinds[nreq] = 2; levels[nreq++] = 6; kinds[nreq] = 2; levels[nreq++] = 5; }
    if (do_local) { kinds[nreq] = 2; levels[nreq++] = 3; }
    EGONN_TRY(ensure_rowgroups(c, kinds, levels, nreq, st));
    DBG_SYNC("row groups");
  }

  // ---- workspace: every intermediate gets its own buffer (HBM is plentiful; no aliasing hazards)
  size_t need = (size_t)P.cap[0] * (1 + 32) * 4;
  for (int i = 1; i <= 7; ++i) need += (size_t)P.cap[i] * 128 * 4 * 6 + (size_t)P.lv[i].rg27.cap_groups * 128 * 4;
  need += (size_t)P.cap[5] * (192 + 256 + 128 * 2) * 4 + (size_t)P.cap[3] * (96 + 32 + 32 + 3 + 64 * 2) * 4;
  need += (size_t)B * SEG_CHUNKS * 256 * 4 * 10 + (size_t)P.cap[3] * 32 + (4u << 20);
  const size_t ks_floats = sconv_ksplit_scratch_floats(c);
  need += ks_floats * 4 + 4096;
  for (int l = 0; l < EGONN_NUM_LEVELS; ++l) c->level_feat[l] = nullptr;
  EGONN_TRY(c->work_arena.ensure(need));
  Arena& A = c->work_arena;
  A.reset();
  // partial tiles of the offset-split launches: ONE buffer, written by a convolution and consumed by its reducer before the
  // next launch of the same stream writes it again
  c->ks_part = ks_floats ? A.alloc<float>(ks_floats) : nullptr;
  c->ks_part_floats = c->ks_part ? ks_floats : 0;
#define WALLOC(var, count)                                                         \
  float* var = A.alloc<float>((size_t)(count));                                    \
  EGONN_REQUIRE(var != nullptr, EGONN_ERR_STATE, "work arena too small (" #var ")")
  // feature maps: `count` elements of the map precision (bf16 maps use half of the fp32-sized slot)
#define FALLOC(var, count)                                                         \
  void* var = A.alloc<char>((size_t)(count) * es);                                 \
  EGONN_REQUIRE(var != nullptr, EGONN_ERR_STATE, "work arena too small (" #var ")")

  // ---- trunk (models/minkgl.py:136-153)
  const int64_t n0 = P.cap[0];
  const float* f0 = features;          // voxelize plans: features are already in level-0 row order; NULL = all ones
  if (features && !c->from_points) {
    WALLOC(fg, n0);
    EGONN_TRY(gather_rows(features, P.perm0, n0, 1, fg, st, cnt));
    f0 = fg;
  }
  FALLOC(x0, n0 * 32);
  {
    ProfScope ps(c, st, "conv0_k5_kernel/L0", PK_CONV0, 0, 125, 1, 32, (int)es);
    EGONN_TRY(conv0_k5_forward(c, f0, m->conv0, 32, m->bn[0].scale, m->bn[0].shift, 1, x0, bf16, st, m->conv0_unit));
  }
  DBG_SYNC("conv0");
  const void* x[8] = {x0};
  c->level_feat[0] = x0;
  c->level_ch[0] = 32;
  c->level_bf16 = bf16;
  // ---- local head, descriptor / keypoint / sigma regressors (models/minkgl.py:287-308)
  auto local_head = [&](hipStream_t st) -> int {
    const int64_t n3 = P.cap[3], n4 = P.cap[4];
    FALLOC(l4, n4 * LOCAL_CH);
    EGONN_TRY(dense_forward_ex(x[4], bf16, n4, 128, m->l1x1[4], 0, LOCAL_CH, nullptr, nullptr, nullptr, ACT_NONE, nullptr, 0, l4,
                               bf16, st, cnt + 4));
    FALLOC(u3, n3 * LOCAL_CH);
    {
      char tag[64];
      snprintf(tag, sizeof(tag), "%s<%d,%d>/L3/tconv", sconv_kernel_name(c, 2, 3, LOCAL_CH, LOCAL_CH, bf16),
               LOCAL_CH, LOCAL_CH);
      ProfScope ps(c, st, tag, PK_TCONV, 3, 8, LOCAL_CH, LOCAL_CH, (int)es);
      EGONN_TRY(sconv_map(c, 2, 3, l4, nullptr, bf16 ? m->q_lt[4] : m->p_lt[4], m->s_lt[4], LOCAL_CH, LOCAL_CH, bf16, nullptr, nullptr, 0, u3,
                          nullptr, nullptr, 0, st));
    }
    EGONN_REQUIRE(m->ldec.cin == 64 && m->ldec.mid == 96 && m->ldec.cout == 128 && m->kp.mid == 32 && m->sg.mid == 32,
                  EGONN_ERR_STATE, "local heads: unexpected layer sizes");
    const float* hw[12] = {m->ldec.w0, m->ldec.b0, m->ldec.w1, m->ldec.b1, m->kp.w0, m->kp.b0, m->kp.w1, m->kp.b1,
                           m->sg.w0, m->sg.b0, m->sg.w1, m->sg.b1};
    static const bool fuse_lateral = getenv("EGONN_NO_FUSED_LATERAL") == nullptr;      // measurement switch
    static const bool heads_split_ok = getenv("EGONN_NO_SPLIT_HEADS") == nullptr;      // measurement switch
    // the heads' Linear layers on the fp16 matrix pipe (dense.hip), unless this context asked for exact fp32 arithmetic
    const void* lh_pack = (heads_split_ok && m->lh_pack && c->split_max_level >= 0 && c->conv_variant == 0) ? m->lh_pack : nullptr;
    if (fuse_lateral && LOCAL_CH == 64) {
      // the level-3 lateral 1x1 convolution + the transposed convolution's output are the first layer of the heads' kernel — the
      // 64-channel map they read is never written (bitwise the rows of the dense launch this replaces; bf16 maps are widened on load)
      EGONN_TRY(local_heads_forward(reinterpret_cast<const float*>(x[3]), n3, cnt + 3, hw, P.lv[3].keys, 3, P.coord_bits, quant_mode,
                                    step, (flags & EGONN_FLAG_IGNORE_KP_REGRESSOR) ? 1 : 0, out_desc, out_kp, out_sigma, st,
                                    m->l1x1[3], reinterpret_cast<const float*>(u3), bf16, lh_pack, c->dev_flags));
      return EGONN_OK;
    }
    WALLOC(l3, n3 * LOCAL_CH);
    EGONN_TRY(dense_forward_ex(x[3], bf16, n3, 64, m->l1x1[3], 0, LOCAL_CH, nullptr, nullptr, nullptr, ACT_NONE, u3, bf16, l3, 0, st, cnt + 3));
    EGONN_TRY(local_heads_forward(l3, n3, cnt + 3, hw, P.lv[3].keys, 3, P.coord_bits, quant_mode, step,
                                  (flags & EGONN_FLAG_IGNORE_KP_REGRESSOR) ? 1 : 0, out_desc, out_kp, out_sigma, st, nullptr, nullptr, 0,
                                  lh_pack, c->dev_flags));
    return EGONN_OK;
  };
  static const bool presplit_ok = getenv("EGONN_NO_PRESPLIT") == nullptr;     // measurement switch: conv2 splits in its loop
  const float *gated_t2 = nullptr, *gated_res = nullptr, *gated_gate = nullptr;     // level 1's block tail, evaluated by level 2's k=2 conv
  for (int i = 1; i <= 7; ++i) {
    const BlockRef& b = m->blk[i];
    const Level& L = P.lv[i];
    const int64_t n = P.cap[i];
    FALLOC(y, n * b.cin);
    char tag[64];
    {
      snprintf(tag, sizeof(tag), "%s<%d,%d>/L%d/k2s2", sconv_kernel_name(c, 1, i, b.cin, b.cin, bf16), b.cin,
               b.cin, i);
      ProfScope ps(c, st, tag, PK_K2S2, i, 8, b.cin, b.cin, (int)es);
      // (level 2 with a gated input: the block output of level 1 is evaluated on the gathered rows — see below)
      const bool gin = gated_t2 != nullptr && i == 2;
      c->gated_in2 = gin ? gated_res : nullptr;
      c->gated_gate = gin ? gated_gate : nullptr;
      const int rc = sconv_map(c, 1, i, gin ? gated_t2 : x[i - 1], nullptr, bf16 ? m->q_convs[i] : m->p_convs[i], m->s_convs[i], b.cin, b.cin, bf16,
                               m->bn[i].scale, m->bn[i].shift, 1, y, nullptr, nullptr, 0, st);
      c->gated_in2 = nullptr;
      c->gated_gate = nullptr;
      EGONN_TRY(rc);
    }
    DBG_SYNC("L%d k2s2", i);
    // ECABasicBlock (layers/eca_block.py:56-73)
    FALLOC(t1, n * b.cout);
    // conv1's output has ONE reader, conv2: when both run on the split kernel, conv1's epilogue writes it in split form (the
    // fp16 hi | lo operands conv2 would otherwise make of every gathered fragment in its step loop; sconv_split.hip)
    const bool t1_split = presplit_ok && sconv_uses_split(b.cin, b.cout, bf16, i, c->conv_variant, c->split_max_level) &&
                          sconv_uses_split(b.cout, b.cout, bf16, i, c->conv_variant, c->split_max_level);
    {
      snprintf(tag, sizeof(tag), "%s<%d,%d>/L%d/k3.conv1", sconv_kernel_name(c, 0, i, b.cin, b.cout, bf16),
               b.cin, b.cout, i);
      ProfScope ps(c, st, tag, PK_K3, i, 27, b.cin, b.cout, (int)es);
      c->split_io = t1_split ? 2 : 0;
      const int rc = sconv_map(c, 0, i, y, nullptr, bf16 ? m->q_c1[i] : m->p_c1[i], m->s_c1[i], b.cin, b.cout, bf16, b.n1.scale, b.n1.shift, 1, t1,
                               nullptr, nullptr, 0, st);
      c->split_io = 0;
      EGONN_TRY(rc);
    }
    FALLOC(t2, n * b.cout);
    WALLOC(psum, (size_t)L.rg27.cap_groups * b.cout);
    {
      snprintf(tag, sizeof(tag), "%s<%d,%d>/L%d/k3.conv2", sconv_kernel_name(c, 0, i, b.cout, b.cout, bf16),
               b.cout, b.cout, i);
      ProfScope ps(c, st, tag, PK_K3, i, 27, b.cout, b.cout, (int)es);
      c->split_io = t1_split ? 1 : 0;
      const int rc = sconv_map(c, 0, i, t1, nullptr, bf16 ? m->q_c2[i] : m->p_c2[i], m->s_c2[i], b.cout, b.cout, bf16, b.n2.scale, b.n2.shift, 0, t2,
                               psum, nullptr, 0, st);
      c->split_io = 0;
      EGONN_TRY(rc);
    }
    WALLOC(gate, (size_t)B * b.cout);
    DBG_SYNC("L%d convs", i);
    EGONN_TRY(eca_gate_groups(psum, L.rg27, L.boff, B, b.cout, b.eca, b.eca_k, gate, st));
    DBG_SYNC("L%d eca gate", i);
    const void* res = y;
    static const bool gated_ok = getenv("EGONN_NO_GATED_K2S2") == nullptr;        // measurement switch
    if (i == 1 && gated_ok && !c->keep_level_features && !bf16 && !b.down && c->conv_variant == 0 && b.cout == 32 && m->blk[2].cin == 32 &&
        sconv_uses_split(32, 32, 0, 2, c->conv_variant, c->split_max_level, 1)) {
      // level 1's block output has ONE reader, the strided convolution into level 2, which reads every row exactly once: it
      // evaluates relu(t2 * gate[scan] + y) on the rows it gathers (sconv_split_kernel<32,32,...,GATED>) — the 23 MB map is neither
      // written nor read back and the element-wise launch is gone; bitwise the same level-2 input.  egonn_forward_level_features(1)
      // has nothing to return then.
      gated_t2 = reinterpret_cast<const float*>(t2);
      gated_res = reinterpret_cast<const float*>(y);
      gated_gate = gate;
      x[i] = nullptr;
      c->level_feat[i] = nullptr;
      c->level_ch[i] = b.cout;
      continue;
    }
    FALLOC(xo, n * b.cout);
    static const bool fuse_down = getenv("EGONN_NO_FUSED_DOWN") == nullptr;      // measurement switch
    if (b.down && fuse_down && dense_gate_fusable(n, b.cin, b.cout)) {
      // blocks with a 1x1 downsample branch (levels 2 and 4): the branch, its BatchNorm and the gated residual + ReLU of the block's
      // tail in ONE launch — out = relu(t2 * gate[scan] + bn(y @ Wd)); the branch output never goes to memory (bitwise the result
      // of the two launches it replaces)
      EGONN_TRY(dense_forward_ex(y, bf16, n, b.cin, b.down, 0, b.cout, nullptr, b.dn.scale, b.dn.shift, ACT_NONE, t2, bf16, xo,
                                 bf16, st, cnt + i, gate, L.boff, B));
    } else {
      if (b.down) {
        FALLOC(rd, n * b.cout);
        EGONN_TRY(dense_forward_ex(y, bf16, n, b.cin, b.down, 0, b.cout, nullptr, b.dn.scale, b.dn.shift, ACT_NONE, nullptr, 0, rd,
                                   bf16, st, cnt + i));
        res = rd;
      }
      EGONN_TRY(eca_apply_gate(t2, res, gate, L.boff, B, n, b.cout, xo, bf16, st));
    }
    DBG_SYNC("L%d eca apply", i);
    x[i] = xo;
    c->level_feat[i] = xo;
    c->level_ch[i] = b.cout;
    // (measured: forking the local head onto a second stream here — a parallel branch of the captured graph — shortens
    //  one batch by 4 % but costs 28 % throughput with three graphs in flight: multi-branch graphs launch 4x slower and
    //  serialise against each other; the overlap comes from the batches in flight instead)
  }
  if (do_local) EGONN_TRY(local_head(st));
  DBG_SYNC("local head");

  // ---- global head + decoder + GeM (models/minkgl.py:46-60, 207-225; layers/pooling.py:82-86)
  static const bool fuse_ghead = getenv("EGONN_NO_FUSED_GHEAD") == nullptr;      // measurement switch
  void* g5_fused = nullptr;
  if (do_global && fuse_ghead && !bf16 && c->conv_variant == 0 && P.cap[5] < 8192 && P.cap[6] < 8192 && P.cap[7] < 8192) {
    // MinkHead (models/minkgl.py:46-60) in three launches instead of five: the three lateral 1x1 convolutions depend on the trunk
    // only — ONE grouped launch — and every FPN step `tconv(y) + lateral` is the transposed convolution with the lateral as its
    // epilogue residual.  a + b = b + a: bitwise the five-launch result (tools/check_bitwise_switches.py).
    WALLOC(l7, P.cap[7] * GLOBAL_CH);
    WALLOC(l6, P.cap[6] * GLOBAL_CH);
    WALLOC(l5, P.cap[5] * GLOBAL_CH);
    WALLOC(g6f, P.cap[6] * GLOBAL_CH);
    WALLOC(g5f, P.cap[5] * GLOBAL_CH);
    const float* gin[3] = {reinterpret_cast<const float*>(x[7]), reinterpret_cast<const float*>(x[6]), reinterpret_cast<const float*>(x[5])};
    const int64_t gn[3] = {P.cap[7], P.cap[6], P.cap[5]};
    const int32_t* gnd[3] = {cnt + 7, cnt + 6, cnt + 5};
    const float* gw[3] = {m->g1x1[7], m->g1x1[6], m->g1x1[5]};
    float* gout[3] = {l7, l6, l5};
    EGONN_TRY(dense_small_group3(gin, gn, gnd, gw, gout, st));
    for (int lv = 6; lv >= 5; --lv) {
      char tag[64];
      snprintf(tag, sizeof(tag), "%s<%d,%d>/L%d/tconv", sconv_kernel_name(c, 2, lv, GLOBAL_CH, GLOBAL_CH, 0), GLOBAL_CH, GLOBAL_CH, lv);
      ProfScope ps(c, st, tag, PK_TCONV, lv, 8, GLOBAL_CH, GLOBAL_CH, 4);
      c->conv_residual = lv == 6 ? l6 : l5;
      const int rc = sconv_map(c, 2, lv, lv == 6 ? l7 : g6f, nullptr, m->p_gt[lv + 1], m->s_gt[lv + 1], GLOBAL_CH, GLOBAL_CH, 0, nullptr, nullptr, 0,
                               lv == 6 ? g6f : g5f, nullptr, nullptr, 0, st);
      c->conv_residual = nullptr;
      EGONN_TRY(rc);
    }
    g5_fused = g5f;
    DBG_SYNC("global head (fused)");
  }
  if (do_global) {
    const float* g5 = reinterpret_cast<const float*>(g5_fused);
    if (!g5) {
    FALLOC(g7, P.cap[7] * GLOBAL_CH);
    EGONN_TRY(dense_forward_ex(x[7], bf16, P.cap[7], 128, m->g1x1[7], 0, GLOBAL_CH, nullptr, nullptr, nullptr, ACT_NONE, nullptr, 0,
                               g7, bf16, st, cnt + 7));
    FALLOC(u6, P.cap[6] * GLOBAL_CH);
    {
      char tag[64];
      snprintf(tag, sizeof(tag), "%s<%d,%d>/L6/tconv", sconv_kernel_name(c, 2, 6, GLOBAL_CH, GLOBAL_CH, bf16),
               GLOBAL_CH, GLOBAL_CH);
      ProfScope ps(c, st, tag, PK_TCONV, 6, 8, GLOBAL_CH, GLOBAL_CH, (int)es);
      EGONN_TRY(sconv_map(c, 2, 6, g7, nullptr, bf16 ? m->q_gt[7] : m->p_gt[7], m->s_gt[7], GLOBAL_CH, GLOBAL_CH, bf16, nullptr, nullptr, 0, u6,
                          nullptr, nullptr, 0, st));
    }
    FALLOC(g6, P.cap[6] * GLOBAL_CH);
    EGONN_TRY(dense_forward_ex(x[6], bf16, P.cap[6], 128, m->g1x1[6], 0, GLOBAL_CH, nullptr, nullptr, nullptr, ACT_NONE, u6, bf16,
                               g6, bf16, st, cnt + 6));
    FALLOC(u5, P.cap[5] * GLOBAL_CH);
    {
      char tag[64];
      snprintf(tag, sizeof(tag), "%s<%d,%d>/L5/tconv", sconv_kernel_name(c, 2, 5, GLOBAL_CH, GLOBAL_CH, bf16),
               GLOBAL_CH, GLOBAL_CH);
      ProfScope ps(c, st, tag, PK_TCONV, 5, 8, GLOBAL_CH, GLOBAL_CH, (int)es);
      EGONN_TRY(sconv_map(c, 2, 5, g6, nullptr, bf16 ? m->q_gt[6] : m->p_gt[6], m->s_gt[6], GLOBAL_CH, GLOBAL_CH, bf16, nullptr, nullptr, 0, u5,
                          nullptr, nullptr, 0, st));
    }
    WALLOC(g5u, P.cap[5] * GLOBAL_CH);
    EGONN_TRY(dense_forward_ex(x[5], bf16, P.cap[5], 128, m->g1x1[5], 0, GLOBAL_CH, nullptr, nullptr, nullptr, ACT_NONE, u5, bf16,
                               g5u, 0, st, cnt + 5));
    g5 = g5u;
    DBG_SYNC("global head");
    }
    WALLOC(gh, P.cap[5] * m->gdec.mid);
    WALLOC(gd, P.cap[5] * GLOBAL_DIM);
    EGONN_TRY(run_mlp(m->gdec, g5, P.cap[5], ACT_NONE, gh, gd, st, cnt + 5));
    WALLOC(gp, (size_t)B * SEG_CHUNKS * GLOBAL_DIM);
    // global pooling (layers/pooling.py:13-43): GeM (default, :72-86), SPoC = average (:59-69), MAC = max (:46-56)
    if (flags & (EGONN_FLAG_POOL_SPOC | EGONN_FLAG_POOL_MAC)) {
      const int mode = (flags & EGONN_FLAG_POOL_MAC) ? 2 : 0;
      EGONN_TRY(segment_partial_sums(gd, P.lv[5].boff, B, GLOBAL_DIM, mode, nullptr, gp, st));
      EGONN_TRY(pool_finish(gp, P.lv[5].boff, B, GLOBAL_DIM, mode, out_global, st));
    } else {
      EGONN_REQUIRE(m->gem_p, EGONN_ERR_STATE, "forward: GeM pooling needs the tensor 'global_pooling.pooling.p'");
      EGONN_TRY(segment_partial_sums(gd, P.lv[5].boff, B, GLOBAL_DIM, 1, m->gem_p, gp, st));
      EGONN_TRY(gem_finish(gp, P.lv[5].boff, B, GLOBAL_DIM, m->gem_p, out_global, st));
    }
  }

  DBG_SYNC("global decoder + pooling");
#undef WALLOC
#undef FALLOC
  return EGONN_OK;
}

API int egonn_forward_level_features(egonn_ctx* c, int level, float* out, int channels, void* stream) {
  REQUIRE_PLAN(c);
  EGONN_REQUIRE(level >= 0 && level < EGONN_NUM_LEVELS && c->level_feat[level], EGONN_ERR_STATE,
                "no features for level %d (run egonn_forward first; level 1 of fp32 maps is materialised only after "
                "egonn_debug_keep_level_features(ctx, 1))", level);
  EGONN_REQUIRE(channels == c->level_ch[level], EGONN_ERR_INVALID, "level %d has %d channels, caller expects %d", level,
                c->level_ch[level], channels);
  const int64_t cnt = c->plan.lv[level].n * channels;
  if (c->level_bf16) return convert_bf16_to_f32(c->level_feat[level], cnt, out, (hipStream_t)stream);
  HIP_CHECK(hipMemcpyAsync(out, c->level_feat[level], sizeof(float) * cnt, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return EGONN_OK;
}

API int egonn_select_keypoints(egonn_ctx* c, const float* sigma, const float* keypoints, const float* descriptors,
                               int n_k, float* sel_kp, float* sel_desc, int32_t* sel_rows, int32_t* sel_count,
                               void* stream) {
  REQUIRE_PLAN(c);
  EGONN_REQUIRE(sigma && keypoints && descriptors && sel_kp && sel_desc && sel_rows && sel_count, EGONN_ERR_INVALID,
                "select_keypoints: null argument");
  HIP_CHECK(hipSetDevice(c->device));
  return select_topk(sigma, c->plan.lv[3].boff, c->plan.batch, n_k, keypoints, descriptors, LOCAL_DIM, sel_rows, sel_count,
                     sel_kp, sel_desc, (hipStream_t)stream);
}


// torch.topk(sigma, n_k, largest=False) per segment (eval/evaluate.py:359) without a plan: row_offsets DEVICE int32 (B+1)
API int egonn_topk_rows(const float* sigma, const int32_t* row_offsets, int batch_size, int n_k, int32_t* sel_rows,
                        int32_t* sel_count, void* stream) {
  EGONN_REQUIRE(sigma && row_offsets && sel_rows && sel_count && batch_size >= 1, EGONN_ERR_INVALID, "topk_rows: bad argument");
  return select_topk(sigma, row_offsets, batch_size, n_k, nullptr, nullptr, 0, sel_rows, sel_count, nullptr, nullptr,
                     (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------ launch timing (bench.py)
API int egonn_profile_enable(egonn_ctx* c, int mode, const char* filter) {
  EGONN_REQUIRE(c && mode >= 0 && mode <= 3, EGONN_ERR_INVALID, "profile_enable: bad argument");
  c->prof.mode = mode;
  snprintf(c->prof.filter, sizeof(c->prof.filter), "%s", filter ? filter : "");
  return EGONN_OK;
}

// Drains the timing records collected since the last fetch.  [SYNC]  Returns up to `cap` records:
// names (cap x 64 chars), milliseconds, algorithmic bytes (P*Cin*e + N_out*Cout*e + K*Cin*Cout*e + 8*P; e = 4 fp32 / 2 bf16) and flops
// (2*P*Cin*Cout) of every launch; *n = number of records written.
API int egonn_profile_fetch(egonn_ctx* c, int cap, int* n, char* names, float* ms, double* bytes, double* flops,
                            void* stream) {
  EGONN_REQUIRE(c && n && names && ms && bytes && flops, EGONN_ERR_INVALID, "profile_fetch: null argument");
  HIP_CHECK(hipSetDevice(c->device));
  Plan& P = c->plan;
  if (P.valid) {
    if (P.built_reserved) P.exact = false;      // replays rebuilt the plan: read this batch's sizes
    EGONN_TRY(plan_sync(c, (hipStream_t)stream));
    EGONN_TRY(count_map_pairs(c, (hipStream_t)stream));   // [0] = first-layer pairs, [l] = k=3 map of level l
  }
  HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  unsigned long long pairs[16];
  HIP_CHECK(hipMemcpy(pairs, c->dev_pairs, sizeof(pairs), hipMemcpyDeviceToHost));
  int w = 0;
  std::vector<ProfRec> all(c->prof.recs);
  all.insert(all.end(), c->prof.graph_recs.begin(), c->prof.graph_recs.end());   // the last replay's brackets
  const size_t n_plain = c->prof.recs.size();
  size_t ri = 0;
  for (auto& r : all) {
    const bool plain = ri++ < n_plain;
    float t = 0.f;
    if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) {
      (void)hipGetLastError();
      if (plain) { c->prof.pool.push_back(r.e0); c->prof.pool.push_back(r.e1); }
      continue;                                          // an event a replay has not recorded (yet)
    }
    if (w < cap) {
      // output rows and map pairs of the launch, from the plan's true sizes
      const double n_out = (double)P.lv[r.level].n;
      double Pn = 0;
      switch (r.kind) {
        case PK_CONV0: Pn = (double)pairs[0]; break;
        case PK_K3: Pn = (double)pairs[r.level]; break;
        case PK_K2S2: Pn = (double)P.lv[r.level - 1].n; break;      // every input voxel feeds exactly one parent
        case PK_TCONV: Pn = n_out; break;
        default: Pn = n_out; break;
      }
      snprintf(names + (size_t)w * 64, 64, "%s", r.name);
      ms[w] = t;
      bytes[w] = Pn * r.cin * r.es + n_out * r.cout * r.es + (double)r.K * r.cin * r.cout * r.es + 8.0 * Pn;
      flops[w] = 2.0 * Pn * r.cin * r.cout;
      ++w;
    }
    if (plain) { c->prof.pool.push_back(r.e0); c->prof.pool.push_back(r.e1); }
  }
  c->prof.recs.clear();
  *n = w;
  return EGONN_OK;
}


// ------------------------------------------------------------------------------------------ batch-hard triplet loss
API int64_t egonn_triplet_loss_scratch_floats(int n) { return (int64_t)triplet_loss_scratch_floats(n); }

API int egonn_triplet_loss(const float* embeddings, int n, int d, const uint8_t* positives_mask,
                           const uint8_t* negatives_mask, float margin, float* out_stats, int32_t* out_triplets,
                           float* out_grad, float* scratch, void* stream) {
  EGONN_REQUIRE(embeddings && positives_mask && negatives_mask && out_stats && out_triplets && scratch, EGONN_ERR_INVALID,
                "triplet_loss: null argument");
  return triplet_loss_forward(embeddings, n, d, positives_mask, negatives_mask, margin, out_stats, out_triplets, out_grad,
                              scratch, (hipStream_t)stream);
}


// ------------------------------------------------------------------------------------------ local-head losses
API int egonn_nn_search(const float* a, int64_t n, const float* transform, const float* b, int64_t m, float* out_dist,
                        int32_t* out_index, void* stream) {
  EGONN_REQUIRE(a && b && out_dist && out_index, EGONN_ERR_INVALID, "nn_search: null argument");
  return nn_search(a, n, transform, b, m, out_dist, out_index, (hipStream_t)stream);
}
API int egonn_matrix_min(const float* d, int64_t n, int64_t m, float* row_min, int32_t* row_index, float* col_min,
                         int32_t* col_index, void* stream) {
  EGONN_REQUIRE(d && row_min && row_index && col_min && col_index, EGONN_ERR_INVALID, "matrix_min: null argument");
  return matrix_min(d, n, m, row_min, row_index, col_min, col_index, (hipStream_t)stream);
}
API int egonn_softmax_cross_entropy(const float* logits, int64_t n, int64_t m, const int32_t* target, float* out_loss,
                                    int32_t* out_argmax, float* out_dlogits, void* stream) {
  EGONN_REQUIRE(logits && target && out_loss && out_argmax, EGONN_ERR_INVALID, "softmax_cross_entropy: null argument");
  return softmax_ce(logits, n, m, target, out_loss, out_argmax, out_dlogits, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------ training-mode operators
#define REQUIRE_LEVEL(c, level)                                                                              \
  REQUIRE_PLAN(c);                                                                                           \
  HIP_CHECK(hipSetDevice((c)->device));                                                                      \
  EGONN_REQUIRE((level) >= 0 && (level) < EGONN_NUM_LEVELS, EGONN_ERR_INVALID, "level %d out of range", (level))

API int egonn_dense(const float* x, int64_t n, int cin, const float* weight, int weight_out_in, const float* bias, int cout,
                    int act, float* out, void* stream) {
  EGONN_REQUIRE(x && weight && out && n >= 0 && cin >= 1 && cout >= 1 && act >= 0 && act <= 4, EGONN_ERR_INVALID,
                "dense: bad arguments");
  return dense_forward(x, n, cin, weight, weight_out_in ? 1 : 0, cout, bias, nullptr, nullptr, act, nullptr, out,
                       (hipStream_t)stream);
}

API int egonn_dense_backward_weight(const float* a, int ca, const float* b, int cb, int64_t n, float* out, float* scratch,
                                    int64_t scratch_floats, void* stream) {
  EGONN_REQUIRE(a && b && out && ca >= 1 && cb >= 1 && n >= 0, EGONN_ERR_INVALID, "dense_backward_weight: bad arguments");
  return conv_wgrad(a, b, nullptr, n, 1, ca, cb, out, scratch, (size_t)scratch_floats, (hipStream_t)stream);
}

API int egonn_conv_backward_weight(egonn_ctx* c, int level_in, int level_out, int ks, int transposed, const float* in,
                                   int cin, const float* grad_out, int cout, float* grad_kernel, float* scratch,
                                   int64_t scratch_floats, void* stream) {
  REQUIRE_LEVEL(c, level_in);
  EGONN_REQUIRE(level_out >= 0 && level_out < EGONN_NUM_LEVELS, EGONN_ERR_INVALID, "level %d out of range", level_out);
  EGONN_REQUIRE(grad_out && grad_kernel, EGONN_ERR_INVALID, "conv_backward_weight: null argument");
  hipStream_t st = (hipStream_t)stream;
  const Plan& P = c->plan;
  if (ks == 5) {
    EGONN_REQUIRE(level_in == 0 && level_out == 0 && cin == 1 && cout == 32 && !transposed, EGONN_ERR_INVALID,
                  "conv_backward_weight: k=5 is the 1->32 input layer only");
    return conv0_wgrad(c, in, grad_out, grad_kernel, scratch, (size_t)scratch_floats, st);   // in == NULL: all ones
  }
  EGONN_REQUIRE(in, EGONN_ERR_INVALID, "conv_backward_weight: null input");
  // the MFMA channel plans take their (input row, output row) pairs from the row-group form of the map: built here if no forward
  // call did it before, so that the pair source — and with it the fp32 summation order — never depends on the call history
  if ((ks == 3 || ks == 2) && ((cin == 32 && (cout == 32 || cout == 64)) || (cin == 64 && (cout == 64 || cout == 128)) ||
                               (cin == 128 && cout == 128))) {
    const int kind = ks == 3 ? 0 : (transposed ? 2 : 1);
    if (level_out >= (kind == 2 ? 0 : 1) && level_out < EGONN_NUM_LEVELS - (kind == 2 ? 1 : 0))
      EGONN_TRY(ensure_rowgroups(c, &kind, &level_out, 1, st));
  }
  if (ks == 1) {
    EGONN_REQUIRE(level_in == level_out && !transposed, EGONN_ERR_INVALID, "1x1 conv cannot change the level");
    return conv_wgrad(in, grad_out, nullptr, P.lv[level_in].n, 1, cin, cout, grad_kernel, scratch, (size_t)scratch_floats, st);
  }
  if (ks == 3) {
    EGONN_REQUIRE(level_in == level_out && level_in >= 1 && !transposed, EGONN_ERR_INVALID,
                  "k=3 convolution: levels 1..7, same in/out level");
    return conv_wgrad(in, grad_out, P.lv[level_in].nbr27, P.lv[level_in].n, 27, cin, cout, grad_kernel, scratch,
                      (size_t)scratch_floats, st, &P.lv[level_in].rg27);
  }
  if (ks == 2 && !transposed) {
    EGONN_REQUIRE(level_out == level_in + 1, EGONN_ERR_INVALID, "k=2,s=2 convolution maps level l to l+1");
    return conv_wgrad(in, grad_out, P.lv[level_out].nbr8, P.lv[level_out].n, 8, cin, cout, grad_kernel, scratch,
                      (size_t)scratch_floats, st, &P.lv[level_out].rg8);
  }
  if (ks == 2 && transposed) {
    EGONN_REQUIRE(level_out == level_in - 1 && level_out >= 0, EGONN_ERR_INVALID, "transposed conv maps level l to l-1");
    if (level_out == 0) EGONN_TRY(ensure_level0_parent_table(c, st));
    return conv_wgrad(in, grad_out, P.lv[level_out].nbrT, P.lv[level_out].n, 8, cin, cout, grad_kernel, scratch,
                      (size_t)scratch_floats, st, &P.lv[level_out].rgT);
  }
  set_error("conv_backward_weight: kernel_size %d not supported", ks);
  return EGONN_ERR_INVALID;
}

API int egonn_col_stats(int mode, const float* a, const float* b, const float* mask, const float* mean, int64_t n, int c,
                        float* out, float* scratch, int64_t scratch_floats, void* stream) {
  return col_stats(mode, a, b, mask, mean, n, c, out, scratch, (size_t)scratch_floats, (hipStream_t)stream);
}
API int egonn_bn_train_finalize(const float* sums, const float* shift_point, double count, int c, const float* weight,
                                const float* bias, float eps, float momentum, float* running_mean, float* running_var,
                                float* out_mean_invstd_scale_shift, void* stream) {
  EGONN_REQUIRE(sums && shift_point && weight && bias && out_mean_invstd_scale_shift && count >= 1.0 && c >= 1,
                EGONN_ERR_INVALID, "bn_train_finalize: bad arguments");
  return bn_fwd_finalize(sums, shift_point, count, c, weight, bias, eps, momentum, running_mean, running_var,
                         out_mean_invstd_scale_shift, (hipStream_t)stream);
}
API int egonn_bn_backward_finalize(const float* local_sums, const float* global_sums, double count, int c,
                                   const float* weight, const float* mean, const float* invstd, float* out_abc_dgamma_dbeta,
                                   void* stream) {
  EGONN_REQUIRE(local_sums && global_sums && weight && mean && invstd && out_abc_dgamma_dbeta && count >= 1.0,
                EGONN_ERR_INVALID, "bn_backward_finalize: bad arguments");
  return bn_bwd_finalize(local_sums, global_sums, count, c, weight, mean, invstd, out_abc_dgamma_dbeta, (hipStream_t)stream);
}
API int egonn_affine_act(const float* x, const float* scale, const float* shift, int64_t n, int c, int relu, float* out,
                         void* stream) {
  EGONN_REQUIRE(x && scale && shift && out, EGONN_ERR_INVALID, "affine_act: null argument");
  return affine_act(x, scale, shift, n, c, relu, out, (hipStream_t)stream);
}
API int egonn_affine3(const float* g, const float* mask, const float* x, const float* A, const float* B, const float* C,
                      int64_t n, int c, float* out, void* stream) {
  EGONN_REQUIRE(g && x && A && B && C && out, EGONN_ERR_INVALID, "affine3: null argument");
  return affine3(g, mask, x, A, B, C, n, c, out, (hipStream_t)stream);
}
API int egonn_relu_backward(const float* grad_out, const float* out, int64_t n, int c, float* grad_in, void* stream) {
  EGONN_REQUIRE(grad_out && out && grad_in, EGONN_ERR_INVALID, "relu_backward: null argument");
  return gate_residual_backward(grad_out, out, nullptr, nullptr, 0, n, c, grad_in, nullptr, (hipStream_t)stream);
}
API int egonn_eca_gate(const float* mean, const float* conv_weight, int kernel_size, int batch_size, int channels,
                       float* gate, void* stream) {
  EGONN_REQUIRE(mean && conv_weight && gate, EGONN_ERR_INVALID, "eca_gate: null argument");
  return eca_gate_forward(mean, conv_weight, kernel_size, batch_size, channels, gate, (hipStream_t)stream);
}
API int egonn_eca_gate_backward(const float* grad_gate, const float* gate, const float* mean, const float* conv_weight,
                                int kernel_size, int batch_size, int channels, float* grad_mean, float* grad_weight,
                                void* stream) {
  EGONN_REQUIRE(grad_gate && gate && mean && conv_weight && grad_mean && grad_weight, EGONN_ERR_INVALID,
                "eca_gate_backward: null argument");
  return eca_gate_backward(grad_gate, gate, mean, conv_weight, kernel_size, batch_size, channels, grad_mean, grad_weight,
                           (hipStream_t)stream);
}
API int egonn_act_backward(int act, const float* grad_out, const float* out, int64_t n, int c, float* grad_in, void* stream) {
  EGONN_REQUIRE(grad_out && out && grad_in && act >= 0 && act <= 4, EGONN_ERR_INVALID, "act_backward: bad arguments");
  return act_backward(act, grad_out, out, n, c, grad_in, (hipStream_t)stream);
}
API int egonn_l2_normalize(const float* x, const float* grad_out, int64_t n, int c, float* out, void* stream) {
  EGONN_REQUIRE(x && out && c >= 1, EGONN_ERR_INVALID, "l2_normalize: bad arguments");
  return l2norm_rows(x, grad_out, n, c, out, (hipStream_t)stream);
}
API int egonn_gate_residual(egonn_ctx* c, int level, const float* x, const float* gate, const float* residual, int ch,
                            int relu, float* out, void* stream) {
  REQUIRE_LEVEL(c, level);
  EGONN_REQUIRE(x && out, EGONN_ERR_INVALID, "gate_residual: null argument");
  return gate_residual_forward(x, gate, residual, c->plan.lv[level].boff, c->plan.batch, c->plan.lv[level].n, ch, relu, out,
                               (hipStream_t)stream);
}
API int egonn_gate_residual_backward(egonn_ctx* c, int level, const float* grad_out, const float* out, const float* gate,
                                     int ch, float* grad_x, float* grad_residual, void* stream) {
  REQUIRE_LEVEL(c, level);
  EGONN_REQUIRE(grad_out && grad_x, EGONN_ERR_INVALID, "gate_residual_backward: null argument");
  return gate_residual_backward(grad_out, out, gate, c->plan.lv[level].boff, c->plan.batch, c->plan.lv[level].n, ch, grad_x,
                                grad_residual, (hipStream_t)stream);
}
API int egonn_segment_sums(egonn_ctx* c, int level, int mode, const float* a, const float* b, const float* x2, const float* p,
                           int ch, float* out, float* scratch, int64_t scratch_floats, void* stream) {
  REQUIRE_LEVEL(c, level);
  EGONN_REQUIRE(a && out && mode >= 0 && mode <= 2 && (mode == 1 || b) && (mode != 2 || x2) && (mode != 1 || p),
                EGONN_ERR_INVALID, "segment_sums: bad arguments");
  return seg_sums2(mode, a, b, x2, p, c->plan.lv[level].boff, c->plan.batch, ch, out, scratch, (size_t)scratch_floats,
                   (hipStream_t)stream);
}
API int egonn_segment_broadcast(egonn_ctx* c, int level, const float* v, int ch, int mean, float* out, void* stream) {
  REQUIRE_LEVEL(c, level);
  EGONN_REQUIRE(v && out, EGONN_ERR_INVALID, "segment_broadcast: null argument");
  return seg_broadcast(v, c->plan.lv[level].boff, c->plan.batch, c->plan.lv[level].n, ch, mean, out, (hipStream_t)stream);
}
API int egonn_gem_backward(egonn_ctx* c, int level, const float* x, const float* coef, const float* p, int ch, float* grad_x,
                           void* stream) {
  REQUIRE_LEVEL(c, level);
  EGONN_REQUIRE(x && coef && p && grad_x, EGONN_ERR_INVALID, "gem_backward: null argument");
  return gem_backward_rows(x, coef, p, c->plan.lv[level].boff, c->plan.batch, c->plan.lv[level].n, ch, grad_x,
                           (hipStream_t)stream);
}


// ------------------------------------------------------------------------------------------ retrieval (eval/evaluate.py)
API int egonn_knn(const float* query, int64_t n_query, const float* database, int64_t n_database, int dim, int k,
                  int32_t* out_index, float* out_distance, float* scratch, int64_t scratch_floats, void* stream) {
  EGONN_REQUIRE(n_query < (1ll << 31) && n_database < (1ll << 31), EGONN_ERR_INVALID, "knn: too many rows");
  return knn_search(query, (int32_t)n_query, database, (int32_t)n_database, dim, k, out_index, out_distance, scratch,
                    (size_t)scratch_floats, (hipStream_t)stream);
}
API int egonn_recall_counts(const int32_t* nn_index, const float* query_positions, const float* map_positions,
                            int64_t n_query, int k, int position_dim, const float* radius, int n_radius,
                            int32_t* out_true_positives, void* stream) {
  return recall_counts(nn_index, query_positions, map_positions, (int32_t)n_query, k, position_dim, radius, n_radius,
                       out_true_positives, (hipStream_t)stream);
}


// ------------------------------------------------------------------------------------------ scan ingest
API int64_t egonn_filter_points_scratch_ints(int64_t n) { return (int64_t)ingest_scratch_ints(n); }
API int egonn_filter_points(const float* raw, int64_t n, int floats_per_point, const int64_t* scan_offsets, int batch_size,
                            int remove_zero_points, int remove_ground_plane, float ground_plane_level, float* out_points,
                            int64_t* out_scan_offsets, int32_t* scratch, int64_t scratch_ints, void* stream) {
  return ingest_filter(raw, n, floats_per_point, scan_offsets, batch_size, remove_zero_points, remove_ground_plane,
                       ground_plane_level, out_points, out_scan_offsets, scratch, (size_t)scratch_ints,
                       (hipStream_t)stream);
}
