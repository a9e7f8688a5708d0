/*
 * pocketflow_hip.h -- C ABI of the MI355X (gfx950) kernel library for the PocketFlow
 * compression-training hot path.
 *
 * The reference (Tencent/PocketFlow) has NO native boundary: every "kernel" is a chain of stock
 * TensorFlow ops stitched into the graph from Python.  This header is therefore the boundary a
 * maintainer would bind (ctypes / cffi / pybind) in place of those op chains; each entry point
 * cites the reference op chain (file:line under /root/reference) it replaces.  See INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / TF types.
 *   - every pointer is a DEVICE pointer unless the name ends in _host.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - functions never allocate, never synchronise, never own memory; they enqueue kernels and
 *     return 0 (hipSuccess) or a hipError_t value.  pf_error_string() names it.
 *   - dtype codes: PF_F32 = 0, PF_BF16 = 1 (storage type; arithmetic is always float32).
 *   - activation codes: PF_ACT_NONE = 0, PF_ACT_RELU = 1, PF_ACT_RELU6 = 2.
 *   - weight layout: KRSC = [cout][kh][kw][cin] (dense: [out][in]); the reference's HWIO flat
 *     index (needed by 'split' buckets) is recovered by index arithmetic, never by a transpose.
 *   - activation layout: NHWC, i.e. a [rows = N*H*W][C] row-major matrix.
 *   - min/max slots are pairs of uint32 holding an order-preserving encoding of a float
 *     ({enc(min), ~enc(max)}), initialised to 0xFFFFFFFF (one memset) so that both are updated with
 *     atomicMin and the result is independent of the order of arrival (bit-deterministic).
 */
#ifndef POCKETFLOW_HIP_H_
#define POCKETFLOW_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { PF_F32 = 0, PF_BF16 = 1 };
enum { PF_ACT_NONE = 0, PF_ACT_RELU = 1, PF_ACT_RELU6 = 2 };
enum { PF_BUCKET_TENSOR = 0, PF_BUCKET_CHANNEL = 1, PF_BUCKET_SPLIT = 2 };

/* One weight tensor inside the flat parameter buffer (64 bytes, mirrored by ctypes/numpy). */
typedef struct PfSeg {
  int64_t offset;       /* first element in the flat buffers                                  */
  int64_t len;          /* RS*I*O                                                             */
  int32_t RS;           /* kh*kw of the reference HWIO = [kh][kw][I][O] kernel (dense: 1)     */
  int32_t layout;       /* storage: 0 = KRSC [O][kh][kw][I] (dense [O][I]); 1 = CRS depthwise */
  int32_t I, O;         /* reference cin / cout (depthwise: I = C, O = channel multiplier 1)  */
  int32_t mode;         /* PF_BUCKET_*                                                        */
  int32_t bits;         /* quantisation bits of this tensor (fed per step in the reference)   */
  int32_t bucket_size;  /* 'split' buckets only                                               */
  int32_t n_bucket;     /* number of (alpha,beta) pairs: 1 | O | ceil(len/bucket_size)        */
  int64_t slot_offset;  /* first min/max slot (pair index) of this tensor                     */
  int64_t cb_offset;    /* NUQ: first codebook float; layout [k][n_bucket]                    */
} PfSeg;

/* block -> (segment, chunk) map entry; built once per model on the host.  For the apply kernels in
 * channel mode, row0 / nrows are the first output channel and the number of output channels the
 * chunk touches (their alpha/beta are staged in LDS); 0 / 1 otherwise.                        */
typedef struct PfBlock { int32_t seg; int32_t chunk; int32_t row0; int32_t nrows; } PfBlock;

#define PF_CHUNK 4096   /* elements per block in the segment kernels (256 threads x 16) */

const char* pf_error_string(int err);
int pf_version(void);
/* The launchers' PF_* tuning / A-B environment switches are read once, on first use (csrc/pf_api.hip, struct PfTuning); a tool or
 * test that changes one in-process calls this to have them read again.  No reference counterpart (tuning service). */
int pf_tuning_reload(void);

/* ---- K1: min/max calibration -------------------------------------------------------------
 * replaces tf.reduce_max / tf.reduce_min (+stop_gradient) of __scale,
 * learners/uniform_quantization/utils.py:224-225 == learners/nonuniform_quantization/utils.py:411-412 */
int pf_minmax_slots_init(uint32_t* slots, int64_t n_pairs, void* stream);
/* whole-tensor min/max of act(x); one pass, wave-shuffle + LDS block reduce, 2 atomics / block */
int pf_minmax_tensor(const void* x, int64_t n, int dtype, int act, uint32_t* slot, void* stream);
/* decode slots -> alpha = max - min + 1e-10, beta = min (float pairs), for logging / tests */
int pf_minmax_decode(const uint32_t* slots, int64_t n_pairs, float* alpha_beta, void* stream);

/* ---- K2/K4: uniform fake-quant, forward ----------------------------------------------------
 * replaces (x-beta)/alpha -> *k -> round -> /k -> alpha*.+beta of __uniform_quantize,
 * learners/uniform_quantization/utils.py:163-199, 228-245, and the activation variant
 * insert_quant_op_for_activations :51-79 (act(x) recomputed, per-tensor min/max).            */
int pf_uq_apply(const void* x, void* y, int64_t n, int in_dtype, int out_dtype, int act,
                const uint32_t* slot, int bits, void* stream);
/* backward of act -> fake-quant: STE through round, then ReluGrad / Relu6Grad: dx = g * mask(u) */
int pf_act_grad(const void* g, const void* u, void* dx, int64_t n, int dtype, int act,
                void* stream);

/* ---- K1+K2+K3 over ALL weight tensors of a model in two launches ---------------------------
 * replaces insert_quant_op_for_weights (uq utils.py:81-113) incl. __split_bucket / __channel_bucket
 * (:247-289).  w_flat is the fp32 master buffer, qw_flat the fake-quantised copy the convolutions
 * read (fp32 or bf16).                                                                        */
int pf_seg_minmax(const float* w_flat, const PfSeg* segs, const PfBlock* blocks, int n_blocks,
                  uint32_t* slots, void* stream);
int pf_seg_uq_apply(const float* w_flat, void* qw_flat, int out_dtype, const PfSeg* segs,
                    const PfBlock* blocks, int n_blocks, const uint32_t* slots, void* stream);

/* ---- K5: non-uniform (codebook) fake-quant -------------------------------------------------
 * replaces tile/abs/argmin/gather of __build_norm_quant_point / __build_bucket_norm_quant_point,
 * learners/nonuniform_quantization/utils.py:284-347.  idx (uint8, k <= 256) is kept for backward. */
int pf_seg_nuq_apply(const float* w_flat, void* qw_flat, int out_dtype, uint8_t* idx_flat,
                     const float* codebooks, const PfSeg* segs, const PfBlock* blocks,
                     int n_blocks, const uint32_t* slots, void* stream);
/* dL/dc[j][b] = sum_{i in bucket b, idx_i = j} alpha_b * g_i   (gather-grad scatter-add under the
 * override map {'Mul':'Add','Sign':'Identity'}, nuq utils.py:305-306, 345-346).  The sums are ADDED to dcodebooks
 * [n_codebook] (zeroed by the caller).  Bit-deterministic: terms are accumulated as 2^-36 fixed-point integers with
 * 64-bit integer atomics in acc_ws [n_codebook] (scratch; zeroed here), then converted once.                   */
int pf_seg_nuq_codebook_grad(const void* g_flat, int g_dtype, const uint8_t* idx_flat,
                             float* dcodebooks, int64_t* acc_ws, int64_t n_codebook, const PfSeg* segs,
                             const PfBlock* blocks, int n_blocks, const uint32_t* slots, void* stream);
/* normalised weights x_hat = (w - beta) / alpha of ONE tensor (for the quantile initialiser,
 * nuq utils.py:349-366, which then sorts them).                                               */
int pf_seg_normalize(const float* w_flat, float* xn_out, const PfSeg* segs, int seg_index,
                     const uint32_t* slots, void* stream);

/* ---- K7/K9: weight sparsification ----------------------------------------------------------
 * replaces the prune_op chain of WeightSparseLearner.__build_masks,
 * learners/weight_sparsification/learner.py:283-288.                                          */
int pf_ws_bkup_merge_abs(const float* var, float* bkup, const float* mask, float* abs_out,
                         int64_t n, void* stream);
/* k-th largest of a non-negative float array by 4-pass radix select on the bit pattern
 * (tf.contrib.distributions.percentile = descending sort + index).  workspace: 1024 uint32.   */
int pf_kth_largest_nonneg(const float* a, int64_t n, int64_t k_desc_index, float* out,
                          uint32_t* workspace, void* stream);
int pf_ws_mask_apply(float* var, const float* bkup, float* mask, const float* thr, int64_t n,
                     void* stream);
/* count_nonzero (ws learner.py:51-65) */
int pf_count_nonzero(const float* x, int64_t n, unsigned long long* out, void* stream);

/* ---- K8 (channel pruning): per-channel binary gradient masks -------------------------------
 * replaces mask = ones; mask[:,:,~keep_in,:]=0; mask[:,:,:,~keep_out]=0 ; g*mask of
 * ChannelPrunedLearner.__calc_grads_pruned, learners/channel_pruning/learner.py:406-419.
 * KRSC storage; keep_in[I], keep_out[O] are staged in LDS.                                     */
int pf_cp_build_mask(float* mask, const uint8_t* keep_in, const uint8_t* keep_out, int O, int RS,
                     int I, void* stream);
int pf_cp_mask_grad(float* g, const uint8_t* keep_in, const uint8_t* keep_out, int O, int RS,
                    int I, void* stream);

/* ---- K8+K14: fused (L2-coupled, masked) optimiser steps over a flat buffer ------------------
 * replaces tf.train.AdamOptimizer / MomentumOptimizer + `grad * mask` + the L2 term of calc_loss:
 *   uq learner.py:244-253, ws learner.py:201-212,314-332, cp learner.py:357-368,
 *   nets/resnet_at_ilsvrc12.py:132-135 (loss_w_dcy * sum l2_loss).
 * g_eff = (g * g_scale + wd * p) * mask for the first n_decay elements, (g * g_scale) * mask after;
 * mask may be NULL.  g may be fp32 or bf16.  g_scale = 1/world_size folds the all-reduce average. */
int pf_adam_flat(float* p, const void* g, int g_dtype, float* m, float* v, const float* mask,
                 int64_t n, int64_t n_decay, float wd, float g_scale, float lr, float beta1,
                 float beta2, float eps, float beta1_power, float beta2_power, void* stream);
int pf_momentum_flat(float* p, const void* g, int g_dtype, float* acc, const float* mask,
                     int64_t n, int64_t n_decay, float wd, float g_scale, float lr, float momentum,
                     void* stream);
/* The same two updates for a step that was captured in a hipGraph (pocketflow_amd/step_graph.py; the reference's counterpart is
 * the ONE `sess.run(train_op)` of uq learner.py:172 -- a TF session also replays a compiled graph whose learning rate is a tensor,
 * every learner's `setup_lrn_rate`): the per-step scalars are read from device memory, hp[0] = Adam's
 * alpha_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t) (float32, computed by the host exactly as pf_adam_flat does), hp[1] = lr.
 * pf_set_floats writes dst[0..3] from by-value kernel arguments (no staging buffer the host could overrun). */
int pf_adam_flat_dev(float* p, const void* g, int g_dtype, float* m, float* v, const float* mask,
                     int64_t n, int64_t n_decay, float wd, float g_scale, const float* hp, float beta1,
                     float beta2, float eps, void* stream);
int pf_momentum_flat_dev(float* p, const void* g, int g_dtype, float* acc, const float* mask,
                         int64_t n, int64_t n_decay, float wd, float g_scale, const float* hp,
                         float momentum, void* stream);
int pf_set_floats(float* dst, float a, float b, float c, float d, void* stream);

/* ---- K10+K11: distillation + hard-label losses, forward and backward in one kernel ---------
 * replaces tf.losses.softmax_cross_entropy(labels, logits) (nets/<model>.py calc_loss) and
 * DistillationHelper.calc_loss (learners/distillation_helper.py:86-103).
 *   L_model = mean_n CE(labels_n, z_s_n) ; L_dst = loss_w * mean_n CE(softmax(z_t/T), z_s/T)
 *   dz_s = (softmax(z_s)*sum(labels) - labels)/B + loss_w/(B*T) * (softmax(z_s/T) - softmax(z_t/T))
 * labels: float one-hot / soft [B][C].  z_t may be NULL (no distillation).  losses[0] = L_model,
 * losses[1] = L_dst.  row_ws: B*2 floats of scratch.  One workgroup per row, wave-shuffle
 * reductions, fixed summation order (bit-deterministic).                                      */
int pf_ce_distill_fwd_bwd(const void* z_s, int zs_dtype, const float* labels, const void* z_t,
                          int zt_dtype, int B, int C, float tempr, float loss_w, float* losses,
                          void* dz_s, int dz_dtype, float* row_ws, void* stream);

/* ---- K13 fused with K4: batch-norm (training) + ReLU + activation fake-quant ----------------
 * replaces tf.layers.batch_normalization(fused=True) -> tf.nn.relu -> (min/max, quantise) of
 * utils/external/resnet_model.py:55-62 + uq utils.py:51-79 for the BN->ReLU->conv chains of
 * ResNet-v2 / MobileNet-v1.  x is [rows][C] (NHWC).
 * pass 1 (pf_bn_stats): per-channel sum, sum of squares, min, max of x in ONE read.
 * finalize: mean/var -> scale/shift, moving averages (unbiased variance), and -- because
 *   y = act(scale*x+shift) is monotone in x per channel -- the exact whole-tensor min/max of y
 *   from the per-channel min/max of x, written into the activation's min/max slot.
 * pass 2 (pf_bn_act_quant_apply): read x, write q = fake_quant(act(scale*x+shift)).            */
int pf_bn_stats(const void* x, int dtype, int64_t rows, int C, float* partial /*[nblk][4][C]*/,
                int n_blocks, void* stream);
int pf_bn_finalize(const float* partial, int n_blocks, int64_t rows, int C, const void* x_row0,
                   int dtype, const float* gamma, const float* beta, float* moving_mean,
                   float* moving_var, float momentum, float eps, int training /*0: use moving stats*/,
                   int act, float* scale_shift /*[2][C]*/, float* mean_invstd /*[2][C]*/,
                   uint32_t* slot /*may be NULL*/, void* stream);
int pf_bn_act_quant_apply(const void* x, void* q, int dtype, int64_t rows, int C,
                          const float* scale_shift, int act, const uint32_t* slot, int bits,
                          int quantize, void* stream);
/* backward: dy = dq * actmask(scale*x+shift) (STE); per-channel sum(dy), sum(dy*xhat) */
int pf_bn_bwd_stats(const void* dq, const void* x, int dtype, int64_t rows, int C,
                    const float* scale_shift, const float* mean_invstd, int act,
                    float* partial /*[nblk][2][C]*/, int n_blocks, void* stream);
int pf_bn_bwd_finalize(const float* partial, int n_blocks, int C, float* dgamma, float* dbeta,
                       void* stream);
int pf_bn_bwd_apply(const void* dq, const void* x, void* dx, int dtype, int64_t rows, int C,
                    const float* scale_shift, const float* mean_invstd,
                    const float* dgamma, const float* dbeta, int act, void* stream);
/* inference-mode BN + act (teacher forward, eval graphs): y = act(scale*x+shift) */
/* pf_bn_bwd_apply + the gradient that reaches the same tensor through an identity shortcut:
 * dx = bn_backward(dq, x) + addend  (replaces the AddN of the two gradient paths TF inserts for a tensor
 * with two consumers, utils/external/resnet_model.py:257-314 `inputs + shortcut`).  addend may be NULL. */
int pf_bn_bwd_apply_add(const void* dq, const void* x, const void* addend, void* dx, int dtype,
                        int64_t rows, int C, const float* scale_shift, const float* mean_invstd,
                        const float* dgamma, const float* dbeta, int act, void* stream);
int pf_bn_eval_scale_shift(const float* gamma, const float* beta, const float* moving_mean,
                           const float* moving_var, float eps, int C, float* scale_shift,
                           void* stream);

/* ---- K12 fused with K13/K4: 1x1 convolutions with BN/ReLU/fake-quant prologue and residual-add /
 * BN-statistics epilogue (bf16 NHWC, fp32 accumulate) -------------------------------------------
 * replaces, per bottleneck block, the chain  tf.layers.batch_normalization -> tf.nn.relu ->
 * __uniform_quantize(activation) -> tf.nn.conv2d(1x1) [-> + shortcut] of
 * utils/external/resnet_model.py:257-314 + learners/uniform_quantization/utils.py:51-79,92-103 and the
 * Conv2DBackpropInput / Conv2DBackpropFilter gradients of that convolution.
 *
 * pf_conv1x1_fwd:  Y[m][n] = sum_k Q(X)[row(m)][k] * W[n][k]  (+ R[m][n])
 *   X [rows_in][K], W [N][K], Y / R [M][N], all bf16.  Prologue (scale_shift != NULL):
 *   Q(x) = fake_quant(act(scale[k]*x + shift[k])) with scale_shift = {scale[K], shift[K]} as written by
 *   pf_bn_finalize / pf_bn_eval_scale_shift and the activation range in `slot` (NULL: no fake-quant).
 *   Epilogue: R != NULL adds the residual; partial != NULL receives per-channel {sum, sumsq, min, max} of
 *   the stored Y values as [G][4][N] floats, G = pf_conv1x1_stats_groups_k(M, N, K, scale_shift != NULL), in the layout
 *   pf_bn_finalize consumes (pivot 0).  stride > 1: output pixel (img, ho, wo) of an [.., Ho, Wo] grid
 *   reads input pixel (img, ho*stride, wo*stride) of an [.., H, Wd] grid; ymap != 0 maps the OUTPUT rows
 *   instead (backward-data of a strided conv: X = dY dense, Y = dX pre-zeroed).
 *   Backward-data of a stride-1 conv is the same call with W = the transposed kernel [K][N].
 * pf_conv1x1_wrw:  dW[n][k] = sum_m dY[m][n] * Q(X)[row(m)][k], dW float32 or bf16 [N][K];
 *   workspace: (pf_conv1x1_wrw_splits(M, N, K) + 32) * N * K floats (deterministic staged reduction).
 * Requirements: K % 8 == 0, N % 8 == 0, 16-byte aligned pointers; hipErrorInvalidValue otherwise.     */
int pf_conv1x1_stats_groups(int M, int N);
/* G of the partial-statistics array for an [M][K] x [N][K] problem: shapes whose kernel fits the LDS (K, N <= 512,
 * K * N <= 64 Ki) run on the barrier-free resident-kernel variant (pf_conv_stream.hip), prologue-free shapes with
 * K >= 512 on the direct-to-LDS staged GEMM (pf_igemm.hip); each has its own G.  prologue: scale_shift != NULL.  */
int pf_conv1x1_stats_groups_k(int M, int N, int K, int prologue);
int pf_conv1x1_fwd(const void* X, const void* W, void* Y, const void* R, const float* scale_shift,
                   int act, const uint32_t* slot, int bits, float* partial, int M, int N, int K,
                   int Ho, int Wo, int H, int Wd, int stride, int ymap, void* stream);
/* pf_conv1x1_fwd with the CONSUMER's inference-mode BN + activation folded into the epilogue (round 6):
 *   Y[m][n] = bf16( act_out( out_scale[n] * bf16(conv)[m][n] + out_shift[n] ) ),  out_scale_shift = [2][N] float32
 * -- bit for bit what pf_bn_act_quant_apply(quantize = 0) makes of the stored convolution output, without the write + read of
 * that output in between.  No residual, no statistics.  Used by forward_eval networks: tf.layers.batch_normalization(training=False)
 * behind a convolution (utils/external/resnet_model.py:55-66 under the distillation teacher, learners/distillation_helper.py:60-84).
 * Shapes outside the fused kernels run the plain launch and the stand-alone pass in place (same result). */
int pf_conv1x1_fwd_affine(const void* X, const void* W, void* Y, const float* scale_shift, int act,
                          const float* out_scale_shift, int out_act, int M, int N, int K, int Ho, int Wo, int H, int Wd,
                          int stride, void* stream);
/* backward-data of a stride-1 1x1 convolution, dQ[M][K] = dY[M][N] * W[N][K] (Wt = transposed kernel [K][N]),
 * with the statistics pass of the BN backward of the layer that produced Q fused into the epilogue
 * (replaces FusedBatchNormGrad's reduction over dy, utils/external/resnet_model.py:55-62):
 * partial[G][2][K] = {sum dy, sum dy*xhat}, dy = dQ * act'(scale*x+shift), G = pf_conv1x1_stats_groups_k(M, K, N, 0),
 * in the layout pf_bn_bwd_finalize consumes.                                                              */
int pf_conv1x1_bwd_data_bnstats(const void* dY, const void* Wt, void* dQ, const void* bn_x,
                                const float* bn_scale_shift, const float* bn_mean_invstd, int bn_act,
                                float* partial, int M, int N, int K, void* stream);
int pf_conv1x1_wrw_splits(int M, int N, int K);
int pf_conv1x1_wrw(const void* dY, const void* X, void* dW, int dw_dtype, float* workspace,
                   const float* scale_shift, int act, const uint32_t* slot, int bits, int M, int N,
                   int K, int Ho, int Wo, int H, int Wd, int stride, void* stream);

/* ---- K12: dense RxS convolutions as implicit GEMMs on the matrix cores (pf_igemm.hip) ----------------------------
 * replaces tf.nn.conv2d / Conv2DBackpropInput of the 3x3 convolutions of the ResNet blocks
 * (utils/external/resnet_model.py:92-103 conv2d_fixed_padding, :257-314 conv2 of _bottleneck_block_v2) and serves
 * as the plain GEMM of the 1x1 convolutions that carry no prologue.
 *
 * pf_conv2d_fwd:  Y[img][ho][wo][n] = sum_{r,s,c} X[img][ho*stride + r - pad_h][wo*stride + s - pad_w][c] * W[n][r][s][c]
 *   X [imgs][H][Wd][C] bf16 NHWC, W [N][th][tw][C] bf16 (KRSC), Y [imgs][Ho][Wo][N] bf16; taps outside the image read
 *   zeros (`zero`: >= 128 zero bytes in device memory, 16-byte aligned).  C % 64 == 0, N % 8 == 0.
 *   Epilogue: R != NULL adds a residual [M][N]; partial != NULL receives per-channel {sum, sumsq, min, max} of the stored
 *   values as [G][4][N], G = pf_conv2d_stats_groups_geom(<the arguments of this call>), M = imgs*Ho*Wo (layout of
 *   pf_bn_finalize, pivot 0); with
 *   bn_x != NULL instead the BN-backward sums {sum dy, sum dy*xhat} [G][2][N] of the BN whose input is bn_x [M][N]
 *   (dy = Y * act'(scale*x + shift)), as pf_conv1x1_bwd_data_bnstats.
 *   Backward-data of a stride-1 convolution is the same call on dY with the kernel flipped and transposed:
 *   W'[c][r][s][n] = W[n][th-1-r][tw-1-s][c], pad' = th-1-pad.                                                      */
/* rows G of the statistics array for a 1x1 product of M x N outputs (DEPRECATED for R x S convolutions: the tile, and with it G,
 * depends on the geometry) */
int pf_conv2d_stats_groups(int M, int N);
/* rows G for the pf_conv2d_fwd call with these arguments: the ONLY valid query for R x S convolutions                          */
int pf_conv2d_stats_groups_geom(int imgs, int H, int Wd, int C, int N, int th, int tw, int stride, int pad_h, int pad_w,
                                int Ho, int Wo);
int pf_conv2d_fwd(const void* X, const void* W, void* Y, const void* zero, const void* R, float* partial,
                  const void* bn_x, const float* bn_scale_shift, const float* bn_mean_invstd, int bn_act,
                  int imgs, int H, int Wd, int C, int N, int th, int tw, int stride, int pad_h, int pad_w,
                  int Ho, int Wo, void* stream);
/* pf_conv2d_fwd with the consumer's inference-mode BN + activation folded into the epilogue: see pf_conv1x1_fwd_affine */
int pf_conv2d_fwd_affine(const void* X, const void* W, void* Y, const void* zero, const float* out_scale_shift, int out_act,
                         int imgs, int H, int Wd, int C, int N, int th, int tw, int stride, int pad_h, int pad_w, int Ho, int Wo,
                         void* stream);

/* pf_conv2d_wrw: backward-filter of the same convolutions (Conv2DBackpropFilter), dW[n][r][s][c] = sum_m dY[m][n] *
 * X[pix(m, r, s)][c], dW float32 or bf16 in KRSC layout; deterministic (fixed-order reductions, no atomics).
 * C % 64 == 0, N % 64 == 0, M = imgs*Ho*Wo >= 2048; workspace: (pf_conv2d_wrw_splits(M, N, C, th*tw) + 32) * N*th*tw*C
 * floats (pf_conv2d_wrw_splits returns 0 for unsupported shapes).                                                   */
int pf_conv2d_wrw_splits(int M, int N, int C, int taps);
int pf_conv2d_wrw(const void* dY, const void* X, void* dW, int dw_dtype, float* workspace, int imgs, int H, int Wd, int C,
                  int N, int th, int tw, int stride, int pad_h, int pad_w, int Ho, int Wo, void* stream);

/* ---- backward-data layout of all convolution kernels in one launch -------------------------------------------------
 * Conv2DBackpropInput needs W'[c][R-1-r][S-1-s][n] = W[n][r][s][c]; one 64 x 64 tile of one (kernel, tap) matrix per
 * workgroup, described by a PfTransposeTile (built once on the host).  src / dst: flat buffers of `dtype`.            */
typedef struct PfTransposeTile {
  int64_t src_off, dst_off;   /* element offsets of the [O][I] source matrix / the [I][O] destination matrix        */
  int32_t O, I, src_ld, dst_ld;
  int32_t o0, i0;             /* tile origin                                                                        */
  int32_t reserved0, reserved1;
} PfTransposeTile;
int pf_seg_transpose(const void* src_flat, void* dst_flat, int dtype, const void* tiles, int n_tiles, void* stream);

/* ---- K13: max-pooling of the ResNet stem (tf.layers.max_pooling2d, padding SAME: utils/external/resnet_model.py:522-526)
 * and MaxPoolGrad, NHWC float32 / bf16, C % 8 == 0.  Padding is -inf (clipped windows).  idx[B][Ho][Wo][C] (uint8, may be
 * NULL in the forward call) holds r*k+s of the FIRST maximum of each window, the element the gradient is routed to.
 * pf_maxpool_bwd writes every dx element exactly once (gather over the windows that contain it): no atomics.           */
int pf_maxpool_fwd(const void* x, void* y, void* idx, int dtype, int B, int H, int W, int C, int k, int stride,
                   int pad_h, int pad_w, int Ho, int Wo, void* stream);
int pf_maxpool_bwd(const void* dy, const void* idx, void* dx, int dtype, int B, int H, int W, int C, int k, int stride,
                   int pad_h, int pad_w, int Ho, int Wo, void* stream);

/* ---- the ResNet stem: 7x7 / stride 2 / pad 3 convolution of a 3-channel image to 64 channels --------------------
 * replaces the initial conv2d_fixed_padding of utils/external/resnet_model.py:478-486 (tf.pad + tf.layers.conv2d VALID).
 * X [imgs][H][Wd][3] bf16 (NHWC), W [64][7][7][3] bf16 (KRSC), Y [imgs][H/2][Wd/2][64] bf16.  H even, Wd % 32 == 0:
 * pf_conv_stem_supported tells whether a (H, Wd, C, N, k, stride, pad) convolution is this one.                        */
int pf_conv_stem_supported(int H, int Wd, int C, int N, int k, int stride, int pad);
int pf_conv_stem_fwd(const void* X, const void* W, void* Y, int imgs, int H, int Wd, void* stream);
/* backward-filter of the same convolution (Conv2DBackpropFilter): dW [64][7][7][3] in dw_dtype (PF_F32 / PF_BF16) from
 * dY [imgs][H/2][Wd/2][64] bf16 and X; deterministic (fixed-order slab reduction).  Wd <= 256.  workspace:
 * (pf_conv_stem_wrw_slabs(imgs, H, Wd) + 32) * 64 * 147 floats (pf_conv_stem_wrw_slabs returns 0 for unsupported shapes). */
int pf_conv_stem_wrw_slabs(int imgs, int H, int Wd);
int pf_conv_stem_wrw(const void* dY, const void* X, void* dW, int dw_dtype, float* workspace, int imgs, int H, int Wd,
                     void* stream);

/* ---- the MobileNet-v1 stem (round 5): 3x3 / stride 2 convolution of a 3-channel image to 16 or 32 channels, 'SAME' padding -----
 * replaces the first slim.conv2d of utils/external/mobilenet_v1.py:233-262 (conv_defs[0] = Conv(kernel=[3, 3], stride=2, depth=32);
 * depth multipliers 1.0 and 0.5) and its Conv2DBackpropFilter.  X [imgs][H][Wd][3] bf16 (NHWC), W [N][3][3][3] bf16 (KRSC),
 * Y / dY [imgs][Ho][Wo][N] bf16.  pad_h / pad_w are TensorFlow's FRONT pads (0 | 1); positions behind the image read zeros
 * (no padded copy of the image).  Wd even, Wo % 16 == 0 (pf_conv_stem3_supported).  Backward-filter: dW in dw_dtype, deterministic;
 * workspace (pf_conv_stem3_wrw_slabs(...) + 32) * N * 27 floats (0 slabs: unsupported shape).
 * Pixels must be FINITE: the forward kernel pads its contraction from 27 to 96 with zero weights against real neighbouring pixels, so a
 * NaN / Inf pixel would reach (as 0 * Inf) outputs whose window does not contain it (decoded images are finite).      */
int pf_conv_stem3_supported(int H, int Wd, int C, int N, int k, int stride, int pad_h, int pad_w, int Ho, int Wo);
int pf_conv_stem3_fwd(const void* X, const void* W, void* Y, int imgs, int H, int Wd, int N, int pad_h, int pad_w, int Ho, int Wo,
                      void* stream);
int pf_conv_stem3_wrw_slabs(int imgs, int H, int Wd, int N, int pad_h, int pad_w, int Ho, int Wo);
int pf_conv_stem3_wrw(const void* dY, const void* X, void* dW, int dw_dtype, float* workspace, int imgs, int H, int Wd, int N,
                      int pad_h, int pad_w, int Ho, int Wo, void* stream);

/* ---- K12 (MobileNet): depthwise 3x3 convolutions, NHWC float32 / bf16, kernel [C][3][3] in the activation dtype ----------
 * replaces DepthwiseConv2dNative / ...BackpropInput / ...BackpropFilter behind slim.separable_conv2d(num_outputs=None)
 * (utils/external/mobilenet_v1.py:264-292).  stride 1 | 2; pad_h / pad_w are TensorFlow's FRONT pads of 'SAME'
 * (floor(total / 2); the rest behind is implied by Ho / Wo).  C % 8 == 0 and 256 % (C / 8) == 0 (pf_depthwise_supported).
 * pf_depthwise_fwd: partial (may be NULL) receives the per-channel {sum, sum of squares, min, max} of the stored outputs,
 * [G][4][C] with G = pf_depthwise_groups(B, Ho, Wo, C) -- the statistics of the BatchNorm behind the layer (pf_bn_finalize).
 * pf_depthwise_wrw: slabs = workspace of (pf_depthwise_groups(B, Ho, Wo, C) + 32) * C * 9 floats; fixed-order reduction.  */
int pf_depthwise_supported(int C, int k, int stride);
int pf_depthwise_groups(int B, int Ho, int Wo, int C);
int pf_depthwise_fwd(const void* X, const void* W, void* Y, int dtype, float* partial, int B, int H, int Wd, int C, int k,
                     int stride, int pad_h, int pad_w, int Ho, int Wo, void* stream);
int pf_depthwise_bwd_data(const void* dY, const void* W, void* dX, int dtype, int B, int H, int Wd, int C, int k,
                          int stride, int pad_h, int pad_w, int Ho, int Wo, void* stream);
int pf_depthwise_wrw(const void* dY, const void* X, void* dW, int dtype, int dw_dtype, float* slabs, int B, int H, int Wd,
                     int C, int k, int stride, int pad_h, int pad_w, int Ho, int Wo, void* stream);

/* ---- K13: input pipeline tail -----------------------------------------------------------
 * replaces, per image, the preprocessing chain of utils/external/imagenet_preprocessing.py:226-260 behind the JPEG
 * decoder: training  random_flip_left_right -> tf.image.resize_images(BILINEAR, align_corners=False) -> - means;
 *          eval      _aspect_preserving_resize(256) -> _central_crop(224, 224) -> - means
 * (ilsvrc12_dataset.py:75-93 parse_fn).  `src` holds the decoded uint8 HWC (3-channel) images of one mini-batch back
 * to back; desc[i] describes image i.  Output: [B][OH][OW][3] in out_dtype (PF_F32 / PF_BF16).
 * Resize semantics: TF-1.x legacy bilinear (in = out * scale, no half-pixel offset); the flip mirrors the SOURCE.  */
typedef struct PfImageDesc {
  int64_t offset;          /* first byte of the image inside src                                                  */
  int32_t h, w;            /* source height / width                                                               */
  float scale_y, scale_x;  /* float32(in_size / out_size) of the (virtual) resized image                          */
  int32_t off_y, off_x;    /* top-left corner of the output window inside the resized image (central crop)        */
  int32_t flip;            /* 1: mirror the source horizontally before resizing                                   */
  int32_t reserved;
} PfImageDesc;
int pf_image_resize_bilinear(const void* src, const PfImageDesc* desc, void* out, int out_dtype, int B, int OH,
                             int OW, float mean_r, float mean_g, float mean_b, void* stream);

/* Backward-data of a STRIDED R x S convolution (resnet_model.py:92-103 conv2d_fixed_padding with strides 2; TF's
 * Conv2DBackpropInput) by output-parity classes: stride * stride stride-1 launches of the implicit-GEMM kernel over dY, each walking
 * a sub-grid of the flipped / transposed kernel buffer Wt[C][R][S][N] (Wt[c][r'][s'][n] = W[n][R-1-r'][S-1-s'][c]) in place and
 * scattering its rows to the class's input pixels.  bf16; N % 64 == 0, C % 8 == 0, H % stride == W % stride == 0, R, S >= stride. */
int pf_conv2d_bwd_data_strided(const void* dY, const void* Wt, void* dX, const void* zero, int imgs, int H, int Wd, int C, int N,
                               int R, int S, int stride, int pad_h, int pad_w, int Ho, int Wo, void* stream);
/* ... with the BN-backward sums {sum dy, sum dy * xhat} of the BN whose OUTPUT the convolution read (its input bn_x has dX's shape) in
 * the epilogues, as pf_conv1x1_bwd_data_bnstats: partial [G][2][C], G = pf_conv2d_bwd_data_strided_stats_groups(imgs, H, Wd, C, stride)
 * (one slice of workgroup rows per parity class); replaces a pf_bn_bwd_stats pass over dX and bn_x. */
int pf_conv2d_bwd_data_strided_stats_groups(int imgs, int H, int Wd, int C, int stride);
int pf_conv2d_bwd_data_strided_bnstats(const void* dY, const void* Wt, void* dX, const void* zero, const void* bn_x,
                                       const float* bn_scale_shift, const float* bn_mean_invstd, int bn_act, float* partial,
                                       int imgs, int H, int Wd, int C, int N, int R, int S, int stride, int pad_h, int pad_w,
                                       int Ho, int Wo, void* stream);

/* ---- K12, general form: convolutions of any shape and stride, the dense layer; float32 or bf16 storage, float32 accumulation --
 * replaces tf.nn.conv2d / tf.layers.conv2d / slim.conv2d, their Conv2DBackpropInput / Conv2DBackpropFilter and tf.layers.dense for
 * every shape the MFMA kernels above do not take, and ALL of them in the float32 parity mode:
 *   utils/external/resnet_model.py:92-103 (conv2d_fixed_padding), :552 (dense); utils/external/mobilenet_v1.py:233-292 (slim.conv2d);
 *   nets/lenet_at_cifar10.py:34-68.
 * NHWC activations [imgs][H][W][C], KRSC kernels [N][R][S][C]; pad_h / pad_w are the BEGIN pads (positions outside the image read 0,
 * whatever Ho / Wo say about the end).  A dense layer is the 1x1 convolution of a [imgs][1][1][C] tensor.  Deterministic: one fused
 * multiply-add per term in a fixed order; the backward-filter pixel splits are summed in ascending order.
 * pf_convg_wrw: `slab` = float32 workspace of pf_convg_wrw_splits(...) * N * R * S * C elements; dw [N][R][S][C] in dw_dtype.
 * pf_convg_fwd / _bwd_data: `slab` (optional, may be null) = float32 workspace of slab_elems elements: an output with fewer than 256
 * tiles of 64 x 64 (the dense layer) then splits its contraction over pf_convg_small_splits(M, Nc, K) slabs (0: no split; M = output
 * rows, Nc = output columns, K = contraction length -- a function of the shape alone), summed in ascending order; a workspace
 * smaller than splits * M * Nc elements is hipErrorInvalidValue. */
int pf_convg_small_splits(int M, int Nc, int K);
int pf_convg_fwd(const void* x, const void* w, const float* bias, void* y, int dtype, int imgs, int H, int W, int C, int N, int R,
                 int S, int stride, int pad_h, int pad_w, int Ho, int Wo, float* slab, int64_t slab_elems, void* stream);
int pf_convg_bwd_data(const void* dy, const void* w, void* dx, int dtype, int imgs, int H, int W, int C, int N, int R, int S,
                      int stride, int pad_h, int pad_w, int Ho, int Wo, float* slab, int64_t slab_elems, void* stream);
int pf_convg_wrw_splits(int imgs, int C, int N, int R, int S, int Ho, int Wo);
int pf_convg_wrw(const void* dy, const void* x, void* dw, int dtype, int dw_dtype, float* slab, int imgs, int H, int W, int C, int N,
                 int R, int S, int stride, int pad_h, int pad_w, int Ho, int Wo, void* stream);

/* ---- R x S convolutions with few input channels (ResNet-20 @ CIFAR-10: resnet_model.py:156-199 with 16 / 32 filters) --------------
 * bf16, C % 8 == 0.  pf_im2col gathers Xcol[imgs*Ho*Wo][R*S*C] (taps outside the image: zeros) so that the convolution and its two
 * gradients are 1x1 products on pf_conv1x1_fwd / pf_conv1x1_wrw over Xcol and the [N][R*S*C] view of the KRSC kernel; pf_col2im is
 * the inverse gather of backward-data (per input pixel: float32 sum of its <= R*S terms in tap order, one rounding). */
int pf_im2col(const void* x, void* xcol, int imgs, int H, int W, int C, int R, int S, int stride, int pad_h, int pad_w, int Ho, int Wo,
              void* stream);
int pf_col2im(const void* dxcol, void* dx, int imgs, int H, int W, int C, int R, int S, int stride, int pad_h, int pad_w, int Ho, int Wo,
              void* stream);

/* ---- proximal-gradient channel selection of the 'chn-pruned-gpu' learner ----------------------------------------------------------
 * replaces learners/channel_pruning_gpu/learner.py:379-383 (var_prnd_new = var - lr * grad; var_norm = sqrt(reduce_sum(square, axes
 * [0, 1, 3])); threshold = percentile(var_norm, p); shrk_vec = maximum(1 - threshold / var_norm, 0); assign(var_prnd_new * shrk_vec)).
 * w: float32 master kernel in KRSC storage = [rows = O*R*S][I] with the input channel innermost; g: its gradient (float32 / bf16).
 * pf_prox_norms: norms[I] of W - lr * G (partial: float32 workspace of pf_prox_groups(rows, I) * I elements; deterministic).
 * The threshold is pf_kth_largest_nonneg over `norms` at the nearest-rank index of the percentile (host arithmetic as for K7).
 * pf_prox_apply: W <- (W - lr * G) * max(1 - thr[0] / norms[c], 0). */
int pf_prox_groups(int64_t rows, int I);
int pf_prox_norms(const float* w, const void* g, int g_dtype, float lr, int64_t rows, int I, float* partial, float* norms, void* stream);
int pf_prox_apply(float* w, const void* g, int g_dtype, float lr, int64_t rows, int I, const float* norms, const float* thr, void* stream);

#ifdef __cplusplus
}
#endif
#endif  /* POCKETFLOW_HIP_H_ */
