/* libegonn_hip — C ABI of the MI355X-native EgoNN descriptor-extraction path.
 *
 * Drop-in boundary: the reference (jac99/Egonn) has no native code; its
 * hot path crosses into the third-party MinkowskiEngine 0.5.4 Python bindings.  Each entry point below
 * cites the reference call site (file:line under /root/reference) whose MinkowskiEngine call it
 * replaces.  All pointers are raw device pointers (unless marked HOST), all sizes are explicit, there
 * are no torch types in any signature.  Every call enqueues its work on `stream` (a hipStream_t passed
 * as void*); the only host synchronisations are the size queries marked [SYNC].
 *
 * Conventions
 *   - return value: 0 = ok, non-zero = error; egonn_last_error() returns a thread-local message.
 *   - coordinates: int32 [batch, x, y, z] rows, exactly ME's `batched_coordinates` layout.
 *   - rows of every level are stored in Z-order of (batch, x, y, z): batch-contiguous, deterministic.
 *     (ME's own row order is hash-iteration order and unspecified; consumers may only rely on the
 *     coordinate <-> row association, which egonn_level_coords exposes.)
 *   - levels: level l has tensor stride 2^l (coordinates are multiples of 2^l), l = 0..7.
 *   - fp32 everywhere; sparse-conv kernels are (K, Cin, Cout) / (Cin, Cout), Linear weights (out, in):
 *     the reference's own state_dict layouts.
 */
#ifndef EGONN_HIP_H
#define EGONN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct egonn_ctx egonn_ctx;       /* one per (device, stream user); owns the coordinate plan + workspace */
typedef struct egonn_model egonn_model;   /* EgoNN weights registered by state_dict key */

enum { EGONN_QUANT_CARTESIAN = 0, EGONN_QUANT_POLAR = 1 };
/* status codes returned by every entry point (0 = ok; egonn_last_error() holds the text) */
enum { EGONN_STATUS_OK = 0, EGONN_STATUS_INVALID = 1, EGONN_STATUS_HIP = 2, EGONN_STATUS_RANGE = 3, EGONN_STATUS_STATE = 4,
       EGONN_STATUS_CAPACITY = 5,
       /* an fp32 sparse convolution on the fp16-split matrix pipe met a non-finite accumulator (an activation beyond +-65504,
        * or a non-finite input): the batch's outputs are invalid; egonn_ctx_set_exact_fp32(ctx, 1) and run it again */
       EGONN_STATUS_FP16_RANGE = 6 };
enum { EGONN_FLAG_DISABLE_GLOBAL = 1, EGONN_FLAG_DISABLE_LOCAL = 2, EGONN_FLAG_IGNORE_KP_REGRESSOR = 4,
       /* BASELINE configs[2]: feature maps and sparse-conv weights are bf16 in HBM (2 bytes per element), products
        * accumulate in fp32 on v_mfma_f32_16x16x32_bf16; the dense heads, pooling and all outputs stay fp32 */
       EGONN_FLAG_BF16 = 8,
       /* global pooling of PoolingWrapper (layers/pooling.py:13-43) other than the default GeM: SPoC = average, MAC = max */
       EGONN_FLAG_POOL_SPOC = 16, EGONN_FLAG_POOL_MAC = 32 };

/* ------------------------------------------------------------------ lifecycle / errors */
/* coord_bits in [10,16]: voxel coordinates must lie in [-2^(coord_bits-1), 2^(coord_bits-1)). */
int egonn_ctx_create(egonn_ctx** ctx, int device, int coord_bits);
void egonn_ctx_destroy(egonn_ctx* ctx);
const char* egonn_last_error(void);
/* tests / measurements only: on = 1 routes this context's sparse convolutions through the plain one-thread-per-output
 * HIP kernel (cross-check of the MFMA kernels), on = 2 / 4 force the per-wave / the workgroup-cooperative MFMA kernel (A/B
 * timing); 0 = product choice.  Never set by the product path. */
int egonn_debug_set_naive_conv(egonn_ctx* ctx, int on);
/* tests / measurements only: the offset-split rule of this context's fp32 sparse convolutions on the small maps (the rule is
 * a function of (map class, output level), never of the batch).  map_class 0 = the k=3 maps, 1 = the 8-slot maps (k=2,s=2 and
 * transposed); kparts = offset parts as separate workgroups + a fixed-order reducer launch (1 = none); kw = offset parts inside a
 * workgroup (0 / 1 = none, 2..4); col_parts = column parts per task (0 = automatic); -1 keeps a field.  Every setting sums a
 * row's offsets in a fixed partition and order: results are deterministic and batch-invariant under each, and differ between
 * settings by summation order only (<= 3e-6 of the largest output; tests/test_gpu_ksplit.py). */
int egonn_debug_set_ksplit(egonn_ctx* ctx, int map_class, int level, int kparts, int kw, int col_parts);
/* tests only: on = 1 makes egonn_forward materialise the block output of every level, so that egonn_forward_level_features(ctx, 1, ...)
 * has a map to return (by default level 1's block tail of fp32 maps is evaluated inside level 2's strided convolution and its
 * 23 MB output never exists; bitwise the same results either way) */
int egonn_debug_keep_level_features(egonn_ctx* ctx, int on);
/* measurement hook: buffer for the traced sparse-conv build (debug variant 128): 8 u64 per wave task; NULL = off */
int egonn_debug_set_trace(void* device_buffer);
/* measurement hook: device copies of a map's row-group tables (gmask [groups], snbr [groups][K][16], nullable).  [SYNC] */
int egonn_debug_rowgroup_tables(egonn_ctx* ctx, int map_kind, int level_out, uint32_t* gmask_out, int32_t* snbr_out,
                                int64_t capacity_groups, int64_t* n_groups, void* stream);

/* ------------------------------------------------------------------ coordinate plan
 * replaces ME.utils.sparse_quantize      datasets/quantization.py:42,83   (Cartesian/Polar quantizer __call__)
 *          ME.utils.batched_coordinates  eval/evaluate.py:333, datasets/dataset_utils.py:77
 *          ME.SparseTensor(...)          models/minkgl.py:269             (coordinate-map build)
 *          ME coordinate manager         strided maps + kernel maps of every conv in models/minkgl.py:100-134,39-43
 */
/* Voxelise B scans.  points: (n,3) f32 device, scan b = rows [scan_offsets[b], scan_offsets[b+1]) (HOST, B+1
 * entries).  step: HOST, 1 value (Cartesian) or 3 (polar: degrees, metres, metres).  Builds the plan for the
 * voxelised batch.  [SYNC] */
int egonn_voxelize(egonn_ctx* ctx, const float* points, const int64_t* scan_offsets, int batch_size, int quant_mode,
                   const float* step, void* stream);
/* Capturable plans.  egonn_ctx_reserve fixes the sizes of everything a plan allocates: at most max_points input rows,
 * exactly batch_size scans, at most level_capacity[l] voxels at level l = 0..7 (HOST, 8 entries; NULL = max_points for every
 * level).  After it, egonn_voxelize_device builds the plan WITHOUT any host synchronisation or allocation: points (n_rows,3)
 * f32 device with n_rows <= max_points (rows beyond scan_offsets_dev[B] are ignored), scan_offsets_dev DEVICE int64 (B+1).
 * The per-level row counts stay in device memory, every kernel of egonn_forward / egonn_select_keypoints clips to them,
 * so the sequence voxelize_device -> forward -> select_keypoints can be captured into a hipGraph once and replayed on
 * other batches.  Outputs of a reserved plan are sized by the capacities (out_descriptors: level_capacity[3] rows).
 * egonn_plan_status [SYNC] copies the sizes and the error state of the latest (replayed) plan to the host: non-zero if a
 * coordinate left the +-2^(coord_bits-1) range (status 3) or the batch did not fit the reservation (status 5 =
 * EGONN_STATUS_CAPACITY; the outputs are then invalid — checking the status is MANDATORY for reserved plans: out-of-range /
 * non-finite points are clamped into their sample and overflowing rows are clipped, so an unchecked batch yields
 * plausible-looking but wrong descriptors); afterwards egonn_level_count / egonn_level_batch_offsets return that batch's true sizes. */
int egonn_ctx_reserve(egonn_ctx* ctx, int64_t max_points, int batch_size, const int64_t* level_capacity);
int egonn_voxelize_device(egonn_ctx* ctx, const float* points, int64_t n_rows, const int64_t* scan_offsets_dev,
                          int batch_size, int quant_mode, const float* step, void* stream);
int egonn_plan_status(egonn_ctx* ctx, void* stream);
/* Arithmetic of the fp32 sparse convolutions of this context (models/minkgl.py:105 -> ME's fp32 GEMM).  on = 0 (default): the
 * maps of levels <= 5 run on the fp16 matrix pipe with split operands — x = hi + lo with fp16 parts (|x - hi - lo| <= 2^-22 |x|),
 * weights scaled by a power of two per kernel, the three products hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16 with fp32
 * accumulation: within 3e-6 of the plain fp32 kernel relative to the largest output (tests/test_gpu_graph.py,
 * tests/test_gpu_ksplit.py), the low part of an activation below 2^-3 carrying an ABSOLUTE error <= 2^-25.  RANGE: an fp16 part
 * holds |x| < 65504; a finite activation beyond that turns every accumulator that gathers it into Inf / NaN, which the kernels'
 * epilogues detect before BatchNorm / ReLU can hide it: egonn_plan_status (eager and reserved plans alike) then returns
 * EGONN_STATUS_FP16_RANGE for that batch — there is no silent overflow.  on = 1: every level on the exact fp32 kernels
 * (v_mfma_f32_16x16x4_f32: fp32's range, 1/16 of the matrix rate).  Levels 6-7, bf16 maps and channel plans without a split
 * instantiation always run the exact kernels. */
int egonn_ctx_set_exact_fp32(egonn_ctx* ctx, int on);
/* on = 1: the fp16-split convolutions of this context scale their INPUT map by a power of two per launch (max |in| -> [2^13, 2^14),
 * one reduction launch, undone exactly in the epilogue) — for operands far below 1: the input-gradient convolutions of a training
 * step (training/trainer.py:160-175 -> loss.backward()), whose entries of 1e-6 .. 1e-8 would otherwise lose their fp16 low parts.
 * Eager plans only.  Default 0 (activations of a forward pass sit well inside the range). */
int egonn_ctx_set_operand_autoscale(egonn_ctx* ctx, int on);
/* Row capacity of a level of the current plan (= its row count for eager plans).  No sync. */
int egonn_level_capacity(egonn_ctx* ctx, int level, int64_t* capacity);
/* hipGraph capture of a sequence of calls on `stream` (hipStreamBeginCapture / EndCapture + Instantiate / hipGraphLaunch):
 * for hosts without their own HIP binding.  Everything between begin and end must be capturable: a reserved context that
 * has run the same sequence once (so that no arena grows), no [SYNC] entry point. */
typedef struct egonn_graph egonn_graph;
int egonn_graph_begin(void* stream);
int egonn_graph_end(void* stream, egonn_graph** graph);
int egonn_graph_launch(egonn_graph* graph, void* stream);
void egonn_graph_destroy(egonn_graph* graph);
/* Build the plan from explicit coordinates (N,4) int32 [b,x,y,z] in any row order; duplicate rows collapse onto
 * their first occurrence (ME SparseTensor default).  [SYNC] */
int egonn_coords_set(egonn_ctx* ctx, const int32_t* coords, int64_t n, int batch_size, void* stream);

/* Queries on the current plan (HOST results, no sync: filled by the size query of the call above). */
int egonn_level_count(egonn_ctx* ctx, int level, int64_t* n_rows);
int egonn_level_batch_offsets(egonn_ctx* ctx, int level, int64_t* offsets /* HOST, B+1 */);
/* (N_l,4) int32 coordinates of level `level` in row order. */
int egonn_level_coords(egonn_ctx* ctx, int level, int32_t* out, void* stream);
/* For level-0 row i: index of the caller's point / coordinate row it came from (first occurrence).
 * For egonn_voxelize the index is relative to the scan's own first point, as the reference quantizer returns. */
int egonn_input_index(egonn_ctx* ctx, int64_t* out /* (N0,) device */, void* stream);

/* ------------------------------------------------------------------ operators on the current plan
 * (per-operator entry points; egonn_forward below chains them on device)                                  */
/* MinkowskiConvolution: kernel_size 5 (level_out==level_in==0, Cin=1), 3 (same level), 2 (stride 2,
 * level_out = level_in+1), 1 (dense).  models/minkgl.py:100,105,43,124; ME BasicBlock conv1/conv2.
 * scale/shift (nullable): folded eval-mode MinkowskiBatchNorm; relu: fused MinkowskiReLU. */
int egonn_conv(egonn_ctx* ctx, int level_in, int level_out, int kernel_size, const float* in, int cin,
               const float* kernel, int cout, const float* scale, const float* shift, int relu, float* out,
               void* stream);
/* MinkowskiConvolutionTranspose(k=2,s=2) onto the cached finer map: models/minkgl.py:39,53. level_out = level_in-1 */
int egonn_conv_transpose(egonn_ctx* ctx, int level_in, const float* in, int cin, const float* kernel, int cout,
                         float* out, void* stream);
/* The operator behind the two calls above, with explicit map and precision.  map_kind 0: kernel_size 3 on level_out;
 * 1: kernel_size 2 / stride 2 from level_out-1 into level_out; 2: transposed (k=2,s=2) from level_out+1 onto level_out.
 * bf16 = 1: `in` and `out` are bf16 feature maps (BASELINE configs[2]); the fp32 kernel is rounded to bf16, products
 * accumulate in fp32.  group_sums (nullable): (n_groups, cout) fp32 column sums of the output per group of 16 rows
 * (egonn_map_groups) — the conv2 epilogue form of MinkowskiGlobalPooling (layers/eca_block.py:16,26).  With bf16 maps the
 * sums are taken over the fp32 values BEFORE they are rounded to bf16 for storage (the pooled mean is then the mean of the
 * unrounded activations: closer to the fp32 path than a mean of the stored bf16 numbers; within the configs[2] tolerance).
 * fp32 maps of levels <= 5 run on the fp16 matrix pipe with split operands (fp16 hi + lo parts, three products, fp32
 * accumulate, range guard: see egonn_ctx_set_exact_fp32 above; deviation from the exact fp32 kernel < 3e-6 of the largest
 * output); the maps of levels 3-5 sum a row's offsets in a fixed partition (egonn_debug_set_ksplit).  Non-finite inputs: Inf
 * comes out as NaN / Inf and raises the range flag; NaN stays NaN on both paths. */
int egonn_sparse_conv(egonn_ctx* ctx, int map_kind, int level_out, const void* in, int cin, const float* kernel, int cout,
                      int bf16, const float* scale, const float* shift, int relu, void* out, float* group_sums,
                      void* stream);
/* Builds the row-group tables of every kernel map of the current plan in one launch (otherwise each operator builds the tables
 * of its map on first use): k=3 and k=2,s=2 maps of levels 1..7, transposed maps onto levels 1..6 and, with
 * with_level0_transpose, onto level 0 (input gradient of the first strided convolution, training/trainer.py:168).  No sync. */
int egonn_prepare_maps(egonn_ctx* ctx, int with_level0_transpose, void* stream);
/* n_groups / first_group (HOST, B+1 entries, nullable) of a map's row groups.  [SYNC] */
int egonn_map_groups(egonn_ctx* ctx, int map_kind, int level_out, int64_t* n_groups, int64_t* first_group, void* stream);
/* MinkowskiGlobalAvgPooling: layers/eca_block.py:16, layers/pooling.py:80.  out (B,C) */
int egonn_global_avg_pool(egonn_ctx* ctx, int level, const float* in, int channels, float* out, void* stream);

/* MinkowskiBatchNorm in eval mode (= nn.BatchNorm1d on the rows) folded to per-channel scale/shift for the fused
 * epilogue of egonn_conv:  scale = weight / sqrt(running_var + eps),  shift = bias - running_mean * scale. */
int egonn_bn_fold(const float* weight, const float* bias, const float* running_mean, const float* running_var,
                  float eps, int channels, float* scale, float* shift, void* stream);
/* Tail of a residual block: out = relu(x * gate + residual).  eca_weight (k,) non-NULL: ECALayer gate
 * (MinkowskiGlobalPooling -> Conv1d over channels -> sigmoid -> MinkowskiBroadcastMultiplication,
 * layers/eca_block.py:21-36,66-71); NULL: plain ME BasicBlock tail (out += residual; relu). */
int egonn_block_tail(egonn_ctx* ctx, int level, const float* x, const float* residual, int channels,
                     const float* eca_weight, int eca_ksize, float* out, void* stream);
/* SparseTensor + SparseTensor on the same coordinate map (models/minkfpn.py:91, models/minkgl.py:56). */
int egonn_add(const float* a, const float* b, int64_t n, float* out, void* stream);
/* batch['features'] (caller row order of egonn_coords_set) -> plan row order, (N0, channels). */
int egonn_gather_input(egonn_ctx* ctx, const float* features, int channels, float* out, void* stream);
/* GeM pooling (layers/pooling.py:82-86, third_party/minkloc3d/minkloc.py:47-59): out (B, channels). */
int egonn_gem(egonn_ctx* ctx, int level, const float* x, int channels, const float* p, float* out, void* stream);

/* ------------------------------------------------------------------ model
 * replaces model_factory(...) / MinkGL.forward: models/model_factory.py:31-76, models/minkgl.py:267-315       */
int egonn_model_create(egonn_model** model);
void egonn_model_destroy(egonn_model* model);
/* Register a tensor by its reference state_dict key.  The pointer is borrowed and must
 * stay valid until it is re-registered or the model is destroyed. */
int egonn_model_set_tensor(egonn_model* model, const char* key, const float* data, int ndim, const int64_t* shape);
/* Validate keys/shapes, fold eval-mode BatchNorm into scale/shift.  Call again whenever weights change. */
int egonn_model_finalize(egonn_model* model, void* stream);

/* Forward on the current plan.  features: for an egonn_coords_set plan (n_input, 1) f32 in the CALLER's row
 * order (batch['features'], eval/evaluate.py:334); for an egonn_voxelize plan (N0, 1) in level-0 row order
 * (the voxels did not exist before the call, so there is no caller order).  features == NULL means "all ones" (what
 * the reference always feeds, eval/evaluate.py:334) and selects the occupancy-only first layer.  Outputs (caller-allocated, sizes from egonn_level_count):
 *   out_global (B,256) ; out_descriptors (N3,128) unit-L2 ; out_keypoints (N3,3) metres ; out_sigma (N3,1)
 * rows in level-3 row order (per-sample splits = egonn_level_batch_offsets(3)).  Any output may be NULL when
 * the corresponding head is disabled through `flags`.  quant_mode/step as in egonn_voxelize (needed for
 * Quantizer.keypoint_position, datasets/quantization.py:60-72,93-103).  No host sync. */
int egonn_forward(egonn_ctx* ctx, egonn_model* model, const float* features, int quant_mode, const float* step,
                  int flags, float* out_global, float* out_descriptors, float* out_keypoints, float* out_sigma,
                  void* stream);
/* Feature map of a trunk level produced by the last egonn_forward (debug / parity tests): (N_l, C_l).  Levels whose block output
 * egonn_forward does not materialise return EGONN_ERR_STATE: with fp32 maps level 1's block output is evaluated on the fly by the
 * strided convolution into level 2 (EGONN_NO_GATED_K2S2=1 in the environment materialises it again). */
int egonn_forward_level_features(egonn_ctx* ctx, int level, float* out, int channels, void* stream);

/* Keypoint selection — MinkLocGLEvaluator.get_keypoints_idxes, eval/evaluate.py:352-361: per sample the n_k
 * keypoints with the lowest sigma in increasing order (ties: Z-order of the super-voxel).  Padded outputs:
 *   sel_keypoints (B,n_k,3), sel_descriptors (B,n_k,128), sel_rows (B,n_k) level-3 row or -1, sel_count (B,). */
int egonn_select_keypoints(egonn_ctx* ctx, const float* sigma, const float* keypoints, const float* descriptors,
                           int n_k, float* sel_keypoints, float* sel_descriptors, int32_t* sel_rows,
                           int32_t* sel_count, void* stream);

/* The selection alone on explicit segments: row_offsets DEVICE int32 (batch_size+1), sel_rows (batch_size,n_k) global row or
 * -1, sel_count (batch_size,). */
int egonn_topk_rows(const float* sigma, const int32_t* row_offsets, int batch_size, int n_k, int32_t* sel_rows,
                    int32_t* sel_count, void* stream);

/* ------------------------------------------------------------------ batch-hard triplet loss (training, configs[3])
 * replaces BatchHardTripletLossWithMasks.__call__ (models/loss.py:146-172; miner :114-143; the distance / loss /
 * reducer it calls are pytorch_metric_learning's LpDistance(p=2), TripletMarginLoss(swap=True), AvgNonZeroReducer).
 * embeddings (n,d) f32 — in the sharded step the RCCL all-gathered matrix; masks (n,n) u8.
 * out_stats (10 f32, device): loss, num_triplets, num_non_zero_triplets, avg_embedding_norm, mean/max/min hardest
 * positive distance, mean/max/min hardest negative distance.  out_triplets (n,3) i32: anchor (or -1 if dropped),
 * hardest positive, hardest negative.  out_grad (n,d) nullable: dLoss/dEmbeddings.  scratch: device floats,
 * egonn_triplet_loss_scratch_floats(n) of them.  No host sync. */
int64_t egonn_triplet_loss_scratch_floats(int n);
int egonn_triplet_loss(const float* embeddings, int n, int d, const uint8_t* positives_mask,
                       const uint8_t* negatives_mask, float margin, float* out_stats, int32_t* out_triplets,
                       float* out_grad, float* scratch, void* stream);

/* ------------------------------------------------------------------ local-head losses (training)
 * replaces the dense torch.cdist / torch.min / CrossEntropyLoss work of KeypointLoss (models/loss_utils.py:23-95) and
 * CorrespondenceLoss (:108-139), driven per pair of scans by KeypointCorrLoss (models/loss.py:43-92).  The kernels return
 * indices, distances and d loss / d logits; the differentiable tail on (n,3)/(n,1) tensors is the host's autograd. */
/* Nearest row of b (m,3) for every row of a (n,3), a optionally transformed first by the row-major 4x4 `transform`
 * (misc/poses.py:68-76; device pointer, NULL = none): torch.min(torch.cdist(a', b), dim=1) without the matrix. */
int egonn_nn_search(const float* a, int64_t n, const float* transform, const float* b, int64_t m, float* out_dist,
                    int32_t* out_index, void* stream);
/* torch.min(d, dim=1) and torch.min(d, dim=0) of a dense (n,m) matrix: values and indices (ties: lowest index). */
int egonn_matrix_min(const float* d, int64_t n, int64_t m, float* row_min, int32_t* row_index, float* col_min,
                     int32_t* col_index, void* stream);
/* nn.CrossEntropyLoss rows on (n,m) logits: target (n) int32, < 0 = row ignored.  out_loss (n), out_argmax (n),
 * out_dlogits (n,m) = softmax - onehot (nullable). */
int egonn_softmax_cross_entropy(const float* logits, int64_t n, int64_t m, const int32_t* target, float* out_loss,
                                int32_t* out_argmax, float* out_dlogits, void* stream);

/* ------------------------------------------------------------------ training-mode operators (configs[3])
 * The reference trains through MinkowskiEngine's autograd (training/trainer.py:160-175: model.train(); y = model(batch);
 * loss.backward()).  Backward of a sparse convolution w.r.t. its input is again a sparse convolution on the cached
 * map (k=3: same table with kernel[26-k]^T; k=2,s=2 <-> transposed) and is composed from egonn_conv /
 * egonn_conv_transpose by the host side (egonn_amd/train.py); the entry points below are what has no forward twin.
 * `scratch`: device floats owned by the caller (partial sums of the deterministic two-stage reductions).            */
enum { EGONN_ACT_NONE = 0, EGONN_ACT_RELU = 1, EGONN_ACT_TANH = 2, EGONN_ACT_SOFTPLUS = 3, EGONN_ACT_SIGMOID = 4 };
/* Row-wise dense layer out = act(x @ Wmat + bias): weight_out_in = 0: weight (cin,cout) (ME 1x1 kernel);
 * 1: weight (cout,cin) (nn.Linear / MinkowskiLinear, models/minkgl.py:175-225).  Also the input gradient of either
 * layout (swap the flag). */
int egonn_dense(const float* x, int64_t n, int cin, const float* weight, int weight_out_in, const float* bias, int cout,
                int act, float* out, void* stream);
/* out (ca,cb) = a^T b over n rows: weight gradient of a dense layer (a = input, b = grad_out for the (cin,cout) layout). */
int egonn_dense_backward_weight(const float* a, int ca, const float* b, int cb, int64_t n, float* out, float* scratch,
                                int64_t scratch_floats, void* stream);
/* Weight gradient of MinkowskiConvolution / MinkowskiConvolutionTranspose on the current plan: grad_kernel
 * (K,cin,cout).  kernel_size 5: the 1->32 input layer (in == NULL: all-ones features); 3; 2 (transposed = 0: level
 * l -> l+1, 1: l -> l-1); 1. */
int egonn_conv_backward_weight(egonn_ctx* ctx, int level_in, int level_out, int kernel_size, int transposed,
                               const float* in, int cin, const float* grad_out, int cout, float* grad_kernel,
                               float* scratch, int64_t scratch_floats, void* stream);
/* Per-channel reductions over (n,c) rows -> out (2,c).  MinkowskiBatchNorm in train mode = nn.BatchNorm1d over all
 * rows (models/minkgl.py:102,107):  mode 0: sum a, sum a^2;  mode 1: sum (a-mean)^2, 0;
 * mode 2 (backward): g = a*[mask>0] (mask nullable): sum g, sum g*(b-mean);  mode 3: d = a-mean: sum d, sum d^2 (one-pass
 * statistics around a shift point, additive over ranks for SyncBN).  scratch >= 2*c*max(1024, ceil(n/512)) floats makes the
 * row blocking (and so the fp32 summation order) a function of n only; the minimum accepted is 2*c*ceil(n/512). */
int egonn_col_stats(int mode, const float* a, const float* b, const float* mask, const float* mean, int64_t n, int c,
                    float* out, float* scratch, int64_t scratch_floats, void* stream);
/* Per-channel BatchNorm bookkeeping of nn.BatchNorm1d in train mode, on the device:
 * forward: sums (2,c) from mode 3 around shift_point (c) over `count` rows (whole batch) -> out (4,c) = mean, invstd,
 * scale = weight*invstd, shift = bias - mean*scale; running_mean/var (nullable) updated with `momentum` (unbiased var).
 * backward: local/global (2,c) sums from mode 2 -> out (5,c) = A, B, C of egonn_affine3, dgamma, dbeta. */
int egonn_bn_train_finalize(const float* sums, const float* shift_point, double count, int c, const float* weight,
                            const float* bias, float eps, float momentum, float* running_mean, float* running_var,
                            float* out_mean_invstd_scale_shift, void* stream);
int egonn_bn_backward_finalize(const float* local_sums, const float* global_sums, double count, int c, const float* weight,
                               const float* mean, const float* invstd, float* out_abc_dgamma_dbeta, void* stream);
/* out = relu?(x*scale[c] + shift[c]) — BatchNorm application with batch statistics folded by the caller. */
int egonn_affine_act(const float* x, const float* scale, const float* shift, int64_t n, int c, int relu, float* out,
                     void* stream);
/* out = A[c]*(g*[mask>0]) + B[c]*x + C[c] — BatchNorm input gradient (mask = the ReLU output, nullable). */
int egonn_affine3(const float* g, const float* mask, const float* x, const float* A, const float* B, const float* C,
                  int64_t n, int c, float* out, void* stream);
int egonn_relu_backward(const float* grad_out, const float* out, int64_t n, int c, float* grad_in, void* stream);
/* grad_in = grad_out * act'(.) from the activation's output (MinkowskiReLU / Tanh / Softplus / Sigmoid,
 * models/minkgl.py:181-183,198-200). */
int egonn_act_backward(int act, const float* grad_out, const float* out, int64_t n, int c, float* grad_in, void* stream);
/* MinkowskiFunctional.normalize = F.normalize(x, p=2, dim=1, eps=1e-12) (models/minkgl.py:222-223): grad_out == NULL:
 * out = normalised rows; else out = the input gradient for grad_out. */
int egonn_l2_normalize(const float* x, const float* grad_out, int64_t n, int c, float* out, void* stream);
/* ECALayer gate on the (B, channels) per-sample means: gate = sigmoid(Conv1d(1,1,k,padding=(k-1)/2,bias=False)(mean))
 * (layers/eca_block.py:17-19,28-31) and its backward (grad_mean (B,channels), grad_weight (k,), fixed-order sums). */
int egonn_eca_gate(const float* mean, const float* conv_weight, int kernel_size, int batch_size, int channels, float* gate,
                   void* stream);
int egonn_eca_gate_backward(const float* grad_gate, const float* gate, const float* mean, const float* conv_weight,
                            int kernel_size, int batch_size, int channels, float* grad_mean, float* grad_weight,
                            void* stream);
/* out = relu?(x * gate[sample] + residual): MinkowskiBroadcastMultiplication + residual add + MinkowskiReLU
 * (layers/eca_block.py:69-73) with an explicit (B,c) gate (nullable = 1); backward: d = grad_out*[out>0],
 * grad_residual = d (nullable), grad_x = d*gate. */
int egonn_gate_residual(egonn_ctx* ctx, int level, const float* x, const float* gate, const float* residual, int channels,
                        int relu, float* out, void* stream);
int egonn_gate_residual_backward(egonn_ctx* ctx, int level, const float* grad_out, const float* out, const float* gate,
                                 int channels, float* grad_x, float* grad_residual, void* stream);
/* Per-sample column sums -> out (B,c).  mode 0: a*b;  mode 1: t^p ln t, t = max(a,1e-6) (GeM d/dp);
 * mode 2: (a*[b>0])*x2 (gate gradient).  scratch >= 32*B*c floats. */
int egonn_segment_sums(egonn_ctx* ctx, int level, int mode, const float* a, const float* b, const float* x2,
                       const float* p, int channels, float* out, float* scratch, int64_t scratch_floats, void* stream);
/* out[r] = v[sample(r)] (* 1/n_sample when mean): backward of MinkowskiGlobalPooling (layers/eca_block.py:16). */
int egonn_segment_broadcast(egonn_ctx* ctx, int level, const float* v, int channels, int mean, float* out, void* stream);
/* GeM input gradient: grad_x[r] = coef[sample(r)] * x^(p-1) * [x >= 1e-6] (layers/pooling.py:82-86). */
int egonn_gem_backward(egonn_ctx* ctx, int level, const float* x, const float* coef, const float* p, int channels,
                       float* grad_x, void* stream);

/* ------------------------------------------------------------------ scan ingest (the step before the path)
 * replaces PointCloudLoader.__call__ (misc/point_clouds.py:95-111) after read_pc (datasets/mulran/mulran_raw.py:19-25,
 * datasets/kitti/kitti_raw.py:16-22): raw (n, floats_per_point = 4 | 3) f32 returns of a whole batch (scan b = rows
 * [scan_offsets[b], scan_offsets[b+1]), DEVICE int64, batch_size+1 entries); drops all-zero points (|v| <= 1e-8) and
 * points with z <= ground_plane_level; survivors keep their order.  out_points (n,3) f32 (first out_scan_offsets[B]
 * rows valid), out_scan_offsets (batch_size+1) DEVICE int64.  scratch: egonn_filter_points_scratch_ints(n) int32.
 * n is a CAPACITY (>= scan_offsets[batch_size]; rows beyond scan_offsets[batch_size] are never read), so that a fixed-size
 * launch sequence serves every batch (hipGraph capture).  No host sync: out_scan_offsets can be handed to
 * egonn_voxelize_device as they are (egonn_amd/stream.py), or copied back for egonn_voxelize. */
int64_t egonn_filter_points_scratch_ints(int64_t n);
int egonn_filter_points(const float* raw, int64_t n, int floats_per_point, const int64_t* scan_offsets, int batch_size,
                        int remove_zero_points, int remove_ground_plane, float ground_plane_level, float* out_points,
                        int64_t* out_scan_offsets, int32_t* scratch, int64_t scratch_ints, void* stream);

/* ------------------------------------------------------------------ retrieval (database build, configs[4])
 * replaces the per-query NumPy search of Evaluator.evaluate, eval/evaluate.py:80-82 and :175-176:
 *   embed_dist = np.linalg.norm(map_embeddings - query_embedding, axis=1);  nn_ndx = np.argsort(embed_dist)[:k]
 * out_index (n_query,k) int32 ascending by distance (ties: lower index; -1 beyond n_database), out_distance (n_query,k).
 * scratch >= n_query*n_database floats. */
int egonn_knn(const float* query, int64_t n_query, const float* database, int64_t n_database, int dim, int k,
              int32_t* out_index, float* out_distance, float* scratch, int64_t scratch_floats, void* stream);
/* eval/evaluate.py:84-88 / :181-184: out_true_positives (n_radius,k) int32, [r][nn] = number of queries with a
 * retrieved map element among the first nn+1 whose position is within radius[r] (positions: (n, position_dim) f32). */
int egonn_recall_counts(const int32_t* nn_index, const float* query_positions, const float* map_positions,
                        int64_t n_query, int k, int position_dim, const float* radius, int n_radius,
                        int32_t* out_true_positives, void* stream);

/* ------------------------------------------------------------------ launch timing (bench.py roofline leg)
 * mode 0: off; 1: time every tagged sparse-conv launch (event records around it); 2: only launches whose tag contains
 * `filter`, with the events attached to the kernel dispatch itself (the kernel's own begin..end, also when other streams
 * share the GPU); 3: like 2 but as event-record brackets, which a stream capture turns into graph nodes — the records of
 * a captured sequence are re-recorded by every replay and egonn_profile_fetch returns the LAST replay's durations. */
int egonn_profile_enable(egonn_ctx* ctx, int mode, const char* filter);
/* Drain the records collected since the last fetch.  [SYNC]  names: cap x 64 chars.  bytes = the algorithmic
 * bytes of SURVEY.md §8(d) (P*Cin*4 + N_out*Cout*4 + K*Cin*Cout*4 + 8*P), flops = 2*P*Cin*Cout. */
int egonn_profile_fetch(egonn_ctx* ctx, int cap, int* n, char* names, float* ms, double* bytes, double* flops,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EGONN_HIP_H */
