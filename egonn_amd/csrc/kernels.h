// Internal kernel-launcher declarations shared between the .hip translation units.
#pragma once
#include "common.h"

namespace egonn {

const char* last_error();

// sconv.hip ------------------------------------------------------------------------------------
// out[o] = act( (sum_k in[nbr[o][k]] @ W[k]) * scale + shift ) on the row-group form of a kernel map (rowgroup.hip).
bool sconv_rg_supported(int cin, int cout);
// W [K][cin][cout] (reference layout) -> item-major MFMA fragment order, fp32 or bf16; flip: W'[k] = W[K-1-k];
// transpose: the source kernel is [K][cout][cin]
int pack_rg_weights(const float* W, int K, int cin, int cout, int bf16, int flip, int transpose, void* out,
                    hipStream_t stream);
extern unsigned long long* g_sconv_trace;
// name of the kernel sconv_map dispatches for (map kind, output level, channel plan) under this context's settings
const char* sconv_kernel_name(const Ctx* ctx, int kind, int level, int cin, int cout, int bf16);
// level: output level of the map (selects the prefetch depth / kernel family: a function of the LAYER, never of a capacity)
int sconv_rg_forward(const void* in, int64_t n_in_cap, const RowGroups& rg, int64_t groups_hint, const void* Wp, int cin,
                     int cout, int bf16, const float* scale, const float* shift, int relu, void* out, float* psum,
                     hipStream_t stream, int variant = 0, int level = 0,
                     int split = 0, int32_t* flags = nullptr,    // split: Wp = pack_split_weights form, fp16-split arithmetic (128->128 fp32 maps)
                     const float* residual = nullptr);           // out += residual (fp32 maps) in the epilogue
// three independent (n_i, 128) @ (128, 128) products in one launch (dense.hip)
int dense_small_group3(const float* const* in, const int64_t* n, const int32_t* const* n_dev, const float* const* W, float* const* out,
                       hipStream_t stream);
// Convolution over a map of the plan.  kind 0: k=3 on `level`; 1: k=2,s=2 from level-1 into `level`; 2: transposed from
// level+1 onto `level`.  Wp: kernel already packed for this precision (or null: W is packed into `scratch` first).
// bf16: feature maps in/out and weights are bf16.  psum (nullable): [groups][cout] per-group column sums of the output.
// Wsp: the kernel packed for the split-bf16 path (pack_split_weights; fp32 maps only, nullable like Wp).
int sconv_map(Ctx* ctx, int kind, int level, const void* in, const float* W, const void* Wp, const void* Wsp, int cin, int cout,
              int bf16, const float* scale, const float* shift, int relu, void* out, float* psum, float* scratch,
              size_t scratch_floats, hipStream_t stream);
bool sconv_uses_split(int cin, int cout, int bf16, int level, int variant, int split_max_level, int kind = 0);
// sconv_split.hip: fp32 maps on the fp16 matrix pipe (two-way split operands, three products, fp32 accumulate)
bool sconv_split_supported(int cin, int cout);
int pack_split_weights(const float* W, int K, int cin, int cout, int flip, int transpose, void* out, hipStream_t stream);
size_t split_weights_bytes(int K, int cin, int cout);   // fragments + 16 bytes (inverse of the pack scale)
int sconv_split_default_cfg(int cin, int cout, int64_t groups_hint);
int sconv_split_forward(const float* in, int64_t n_in_cap, const RowGroups& rg, int64_t groups_hint, const void* Wsp, int cin,
                        int cout, const float* scale, const float* shift, int relu, float* out, float* psum, hipStream_t stream,
                        int cfg = 0, int split_io = 0,       // split_io: bit 0 = input map in split form, bit 1 = write split form
                        const float* gated_in2 = nullptr, const float* gated_gate = nullptr, int B = 0,    // input row = relu(in * gate[scan] + in2)
                        int kparts = 1, float* part = nullptr, size_t part_floats = 0,   // offset-split launch: parts, scratch for the partial tiles
                        int col_parts = 0,                                                // column parts per task (0 = automatic)
                        int kw = 0,                                                       // offset parts INSIDE a workgroup (0 / 1: none; 2, 3, 4)
                        int32_t* flags = nullptr,                                         // the plan's flag word (bit 3: fp16 range guard)
                        uint32_t* in_absmax = nullptr, int64_t in_elems = 0,              // operand autoscale: 8 bytes of scratch, elements of `in`
                        const float* residual = nullptr);                                 // out += residual ([n_out][cout] fp32) in the epilogue
size_t sconv_split_part_floats(const RowGroups& rg, int cout, int kparts);
// Offset-split rule of the fp32 lock-step kernels: parts of the map's K offsets (1 = unsplit) and column parts per task for
// (map kind, output level) — a function of the LAYER only (the partition changes the summation order of a row)
void sconv_ksplit_rule(const Ctx* ctx, int kind, int level, int* kparts, int* col_parts, int* kw = nullptr);
void sconv_ksplit_defaults(KsRule* r);                  // the product rule (+ the EGONN_KSPLIT* measurement overrides)
size_t sconv_ksplit_scratch_floats(const Ctx* ctx);    // partial-tile scratch that covers every map of the context's plan
static constexpr size_t SCONV_SCRATCH_FLOATS = (size_t)2 << 20;   // 8 MB: one packed kernel (27 x 256 x 256 fp32 = 7 MB)
// conv.hip -------------------------------------------------------------------------------------
int sconv_naive(const float* in, const int32_t* nbr, const float* W, const float* scale, const float* shift, int relu,
                float* out, int64_t n_out, int K, int cin, int cout, hipStream_t stream);
int conv0_lut_init(Ctx* ctx);      // first-layer lookup table (built once per context)
int conv0_k5_forward(Ctx* ctx, const float* feat, const float* W, int cout, const float* scale,
                     const float* shift, int relu, void* out, int out_bf16, hipStream_t stream, const void* wpk = nullptr);
int conv0_pack_unit(const float* W, void* wpk, hipStream_t stream);   // 24 KB: the unit-feature kernel's W fragments

// dense.hip ------------------------------------------------------------------------------------
enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2, ACT_SOFTPLUS = 3, ACT_SIGMOID = 4 };
// out[r][:] = act( (in[r] @ Wmat + bias) * scale + shift ) (+ residual[r])
//   w_out_in = 0: Wmat = W[cin][cout] (ME 1x1 kernel layout) ; 1: Wmat = W[cout][cin]^T (nn.Linear layout)
int dense_forward(const float* in, int64_t n, int cin, const float* W, int w_out_in, int cout, const float* bias,
                  const float* scale, const float* shift, int act, const float* residual, float* out,
                  hipStream_t stream);
// the same with bf16 feature maps on any of the three row operands (weights and arithmetic stay fp32)
bool dense_gate_fusable(int64_t n, int cin, int cout);
int dense_forward_ex(const void* in, int in_bf16, int64_t n, int cin, const float* W, int w_out_in, int cout,
                     const float* bias, const float* scale, const float* shift, int act, const void* residual, int res_bf16,
                     void* out, int out_bf16, hipStream_t stream, const int32_t* n_dev = nullptr, const float* gate = nullptr,
                     const int32_t* boff = nullptr, int B = 0);   // gate: out = relu(residual * gate[sample][col] + layer output)
int bn_fold(const float* w, const float* b, const float* rm, const float* rv, float eps, int c, float* scale,
            float* shift, hipStream_t stream);
// n_dev (nullable, here and below): device-resident row count (<= n); n then only sizes the grid
int gather_rows(const float* in, const int32_t* perm, int64_t n, int c, float* out, hipStream_t stream,
                const int32_t* n_dev = nullptr);
// per-sample column sums, deterministic two-stage: partial[b][chunk][c]
static constexpr int SEG_CHUNKS = 32;
int segment_partial_sums(const float* in, const int32_t* boff, int B, int c, int pow_mode, const float* p,
                         float* partial, hipStream_t stream);
// out[r] = relu(x[r] * sigmoid(conv1d_k(mean_b))[c] + res[r])  (ECA gate + residual + ReLU)
int eca_apply(const float* x, const float* res, const float* partial, const int32_t* boff, int B, int64_t n, int c,
              const float* wconv, int ksize, float* out, hipStream_t stream);
// ECA gate from the per-group column sums a convolution epilogue left behind (sconv.hip): gate[b][c]
int eca_gate_groups(const float* psum, const RowGroups& rg, const int32_t* boff, int B, int c, const float* wconv, int ksize,
                    float* gate, hipStream_t stream);
// out = relu(x * gate[sample] + res); bf16: x, res and out are bf16 feature maps
int eca_apply_gate(const void* x, const void* res, const float* gate, const int32_t* boff, int B, int64_t n, int c,
                   void* out, int bf16, hipStream_t stream);
int convert_bf16_to_f32(const void* in, int64_t n, float* out, hipStream_t stream);
// GeM: out[b][c] = (mean_b clamp(x,eps)^p)^(1/p) from the pow-mode partial sums
int gem_finish(const float* partial, const int32_t* boff, int B, int c, const float* p, float* out,
               hipStream_t stream);
// SPoC (mode 0: per-sample mean of the sum-mode partials) / MAC (mode 2: per-sample max of the max-mode partials)
int pool_finish(const float* partial, const int32_t* boff, int B, int c, int mode, float* out, hipStream_t stream);
int l2_normalize_rows(float* x, int64_t n, int c, hipStream_t stream, const int32_t* n_dev = nullptr);
int add_act(const float* a, const float* b, int64_t n, int relu, float* out, hipStream_t stream);
// keypoint positions (reference datasets/quantization.py:60-72, 93-103)
int keypoint_positions(const uint64_t* keys, int64_t n, int level, int cb, const float* offsets, int mode,
                       const float* step, int ignore_offsets, float* out, hipStream_t stream, const int32_t* n_dev = nullptr);
// descriptor decoder + L2 norm, keypoint regressor + keypoint_position, sigma regressor (models/minkgl.py:175-225,287-308)
// in one launch on the (n,64) local feature map; w: dw0,db0,dw1,db1, kw0,kb0,kw1,kb1, sw0,sb0,sw1,sb1 (nn.Linear layouts)
int local_heads_forward(const float* x, int64_t n, const int32_t* n_dev, const float* const* w, const uint64_t* keys,
                        int level, int cb, int mode, const float* step, int ignore_offsets, float* out_desc, float* out_kp,
                        float* out_sigma, hipStream_t stream, const float* lateral_w = nullptr, const float* lateral_res = nullptr,
                        int in_bf16 = 0, const void* split_pack = nullptr, int32_t* flags = nullptr);
// the heads' six Linear kernels as fp16 hi | lo fragments (local_heads_split_kernel): w6_dev = DEVICE array of the six weight
// pointers (dw0, dw1, kw0, kw1, sw0, sw1)
size_t local_heads_pack_bytes();
int local_heads_pack(const float* const* w6_dev, void* out, hipStream_t stream);
// top-k smallest sigma per sample, ascending, ties by row (= Z-order) — eval/evaluate.py:352-361 — and the gather of the
// selected keypoints / descriptors, one launch (workgroup = scan); out_kp / out_desc nullable
int select_topk(const float* sigma, const int32_t* boff_dev, int B, int k, const float* kp, const float* desc, int dc,
                int32_t* sel_rows /*[B][k]*/, int32_t* sel_count /*[B]*/, float* out_kp, float* out_desc, hipStream_t stream);

// loss.hip ---------------------------------------------------------------------------------------
size_t triplet_loss_scratch_floats(int n);
int triplet_loss_forward(const float* emb, int n, int d, const uint8_t* pos, const uint8_t* neg, float margin,
                         float* out10, int32_t* triplets, float* grad, float* scratch, hipStream_t stream);

// local-head losses (models/loss_utils.py): searches + softmax cross-entropy rows
int nn_search(const float* a, int64_t n, const float* M, const float* b, int64_t m, float* out_dist, int32_t* out_idx,
              hipStream_t stream);
int matrix_min(const float* d, int64_t n, int64_t m, float* row_min, int32_t* row_idx, float* col_min, int32_t* col_idx,
               hipStream_t stream);
int softmax_ce(const float* logits, int64_t n, int64_t m, const int32_t* target, float* loss, int32_t* argmax, float* dlogits,
               hipStream_t stream);

// train.hip --------------------------------------------------------------------------------------
// dW[k][ci][co] = sum_o in[nbr[o][k]][ci] * dout[o][co]; nbr == nullptr: identity map (K = 1, dense layer)
// rg (nullable): the row-group form of the same map when it is built (the pair source of the MFMA kernel)
int conv_wgrad(const float* in, const float* dout, const int32_t* nbr, int64_t n_out, int K, int cin, int cout,
               float* dW, float* scratch, size_t scratch_floats, hipStream_t stream, const RowGroups* rg = nullptr);
int conv0_wgrad(Ctx* ctx, const float* feat, const float* dout, float* dW, float* scratch, size_t scratch_floats,
                hipStream_t stream);
int col_stats(int mode, const float* a, const float* b, const float* mask, const float* m, int64_t n, int c, float* out2c,
              float* scratch, size_t scratch_floats, hipStream_t stream);
int bn_fwd_finalize(const float* sums, const float* m, double n, int c, const float* w, const float* b, float eps,
                    float momentum, float* running_mean, float* running_var, float* out4, hipStream_t stream);
int bn_bwd_finalize(const float* local, const float* global, double n, int c, const float* w, const float* mean,
                    const float* invstd, float* out5, hipStream_t stream);
int affine_act(const float* x, const float* A, const float* B, int64_t n, int c, int relu, float* out, hipStream_t stream);
int affine3(const float* g, const float* mask, const float* x, const float* A, const float* B, const float* C, int64_t n,
            int c, float* out, hipStream_t stream);
int gate_residual_forward(const float* x, const float* gate, const float* res, const int32_t* boff, int B, int64_t n, int c,
                          int relu, float* out, hipStream_t stream);
int gate_residual_backward(const float* dout, const float* out, const float* gate, const int32_t* boff, int B, int64_t n,
                           int c, float* dx, float* dres, hipStream_t stream);
int seg_broadcast(const float* v, const int32_t* boff, int B, int64_t n, int c, int mean, float* out, hipStream_t stream);
int seg_sums2(int mode, const float* a, const float* b, const float* x2, const float* p, const int32_t* boff, int B, int c,
              float* out_bc, float* scratch, size_t scratch_floats, hipStream_t stream);
int gem_backward_rows(const float* x, const float* coef, const float* p, const int32_t* boff, int B, int64_t n, int c,
                      float* dx, hipStream_t stream);

int eca_gate_forward(const float* mean, const float* w, int ks, int B, int c, float* gate, hipStream_t stream);
int eca_gate_backward(const float* dgate, const float* gate, const float* mean, const float* w, int ks, int B, int c,
                      float* dmean, float* dw, hipStream_t stream);
int act_backward(int act, const float* g, const float* y, int64_t n, int c, float* out, hipStream_t stream);
// g == nullptr: out = normalize(x) ; else out = d normalize / dx applied to g
int l2norm_rows(const float* x, const float* g, int64_t n, int c, float* out, hipStream_t stream);

// retrieval.hip ----------------------------------------------------------------------------------
int knn_search(const float* query, int32_t nq, const float* db, int32_t m, int d, int k, int32_t* out_idx, float* out_dist,
               float* scratch, size_t scratch_floats, hipStream_t stream);
int recall_counts(const int32_t* nn_idx, const float* qpos, const float* mpos, int32_t nq, int k, int pd,
                  const float* radius, int nr, int32_t* tp, hipStream_t stream);

// ingest.hip -------------------------------------------------------------------------------------
size_t ingest_scratch_ints(int64_t n);
int ingest_filter(const float* raw, int64_t n, int stride, const int64_t* raw_off_dev, int batch, int remove_zero,
                  int remove_ground, float ground, float* out_xyz, int64_t* new_off_dev, int32_t* scratch,
                  size_t scratch_ints, hipStream_t stream);

}  // namespace egonn
