/*
 * mimo_hip.h — C ABI of libmimo_hip.so: the MI355X (gfx950) execution engine for the
 * MIMO U-Net forward + backward + loss + optimiser hot path.
 *
 * The reference (antonbaumann/MIMO-Unet @ 2024_10_08) has no FFI seam: the path is plain
 * nn.Module composition.  Each entry point below names the reference interface it
 * replaces (paths relative to the reference root).  All pointers are raw device
 * pointers unless marked "host"; no torch types cross this boundary.  Every function
 * returns 0 on success or a negative mimo_status; mimo_last_error() gives the message
 * (thread-local).  The plan calls (mimo_forward / mimo_loss_forward / mimo_backward* / mimo_adam_step /
 * mimo_uncertainties / the epilogues) are asynchronous on the given hipStream_t, never synchronise and
 * never allocate device memory after mimo_plan_create(); they are not re-entrant on one plan.  The
 * mimo_op_* single-operator entry points at the end exist for the kernel parity tests only: they allocate
 * their own scratch and synchronise the stream before returning.  mimo_plan_profile_read() synchronises on
 * the recorded events.
 */
#ifndef MIMO_HIP_H
#define MIMO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the declarations below are its whole dynamic symbol table
 * (tests/test_host_cpu.py checks `nm -D` against this header). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

typedef struct mimo_plan mimo_plan;
typedef void* mimo_stream; /* hipStream_t */

enum mimo_status {
  MIMO_OK = 0,
  MIMO_ERR_INVALID = -1, /* bad argument / unsupported configuration */
  MIMO_ERR_HIP = -2,     /* a HIP runtime call failed */
  MIMO_ERR_STATE = -3    /* call order violated (e.g. backward before forward) */
};

/* Arithmetic of the 3x3 convolutions (BatchNorm statistics, the loss and the optimiser are always fp32; storage is
 * fp32 except in the two *_MIXED modes).
 * MIMO_PREC_FP32: f32-input MFMA everywhere (bit-exact fp32 fma chains).
 * MIMO_PREC_SPLIT16: each fp32 operand is split into a 16-bit hi + lo pair and every product block of the forward and
 *   the data gradient is three 16-bit MFMAs with fp32 accumulation: fp16 pairs in the forward convolution (~2^-22 per
 *   product: fp32-class outputs; the weight image of a layer carries a power-of-two scale that follows its largest |w|,
 *   so any fp32 weight is in range), bf16 pairs (fp32 exponent range, ~1e-5 per product) in the data gradient.  The
 *   weight gradient (round 5) is two fp16 MFMAs per product — the activation as ONE fp16 value, dz as an fp16 (hi, lo) pair
 *   scaled by a power of two taken from max |dz|: 1-4e-4 of a weight-gradient tensor's scale in rounding noise, nothing
 *   of it reaches other layers (environment MIMO_WGRAD_NP=3: three bf16-pair MFMAs, ~1e-5).
 * MIMO_PREC_BF16: mixed precision in the sense of the reference's `precision="16-mixed"` runs
 *   (scripts/train/train_ndvi.py:71) and of BASELINE config 4: convolution operands are rounded to bf16
 *   on the way into the MFMA (one MFMA per product block, fp32 accumulation); master weights, BatchNorm
 *   statistics, the loss, the optimiser and — in this version — the stored activations stay fp32.
 *   Parity against the fp32 oracle at bf16 tolerance (~1e-2).
 * MIMO_PREC_BF16_MIXED / MIMO_PREC_FP16_MIXED: 16-bit STORAGE as well — activations, convolution outputs and their
 *   gradients live in HBM as bf16 / fp16 (half the bytes of every bandwidth-bound pass), convolution operands are that
 *   type (one MFMA per product, fp32 accumulation); master weights, BatchNorm statistics (taken from the fp32
 *   accumulators), the logits, the loss and the optimiser stay fp32.  FP16_MIXED is the reference's production mode
 *   `precision="16-mixed"` (scripts/train/train_ndvi.py:71): run it under a loss scaler (torch.cuda.amp.GradScaler;
 *   FlatAdam unscales, checks for inf / nan and skips the step on the device).  BASELINE config 4 = BF16_MIXED. */
enum mimo_precision {
  MIMO_PREC_FP32 = 0,
  MIMO_PREC_SPLIT16 = 1,
  MIMO_PREC_BF16 = 2,
  MIMO_PREC_BF16_MIXED = 3,
  MIMO_PREC_FP16_MIXED = 4
};

enum mimo_loss_kind { MIMO_LOSS_LAPLACE_NLL = 0, MIMO_LOSS_GAUSSIAN_NLL = 1 };

/* Block variants: the reference's blocks are BatchNorm2d + ReLU (components.py:22-30) and bilinear
 * align_corners up-sampling (components.py:77-85; its ConvTranspose2d branch, components.py:95-104, is unreachable from
 * the LightningModule: mimo_unet.py:73-74 hard-wire bilinear=True).  Those are the values 0 — the only ones implemented
 * and the only ones that can be parity-checked against the reference; mimo_plan_create rejects any other value with
 * MIMO_ERR_INVALID.  The fields exist so that a GroupNorm / SiLU / transposed-convolution variant (BASELINE.json's
 * north_star wording) is a new enum value, not an ABI change. */
enum mimo_norm_kind { MIMO_NORM_BATCH = 0 };
enum mimo_act_kind { MIMO_ACT_RELU = 0 };
enum mimo_up_kind { MIMO_UP_BILINEAR_ALIGN_CORNERS = 0 };

/* Constructor arguments of mimo.models.mimo_components.model.MimoUNet (model.py:31-44) plus
 * the batch geometry the plan is specialised for.  bilinear=True/use_pooling_indices=False are
 * hard-wired exactly as mimo/models/mimo_unet.py:73-74 hard-wires them. */
typedef struct mimo_config {
  int32_t in_channels;
  int32_t out_channels; /* total head width = 2 * targets (mimo_unet.py:110-111) */
  int32_t num_subnetworks;
  int32_t filter_base_count;
  int32_t batch, height, width;
  float encoder_dropout_rate, core_dropout_rate, decoder_dropout_rate; /* Dropout2d, components.py:29 */
  float bn_eps, bn_momentum; /* 1e-5, 0.1: nn.BatchNorm2d defaults (components.py:24,27) */
  int32_t loss_kind;         /* mimo_loss_kind; mimo/losses.py:39,124 */
  float eps_min, eps_max;    /* clamp of the scale/variance, losses.py:42-45,127-130: 1e-5, 1e3 */
  int32_t device;            /* HIP device ordinal */
  int32_t precision;         /* mimo_precision: arithmetic of the 3x3 forward / data-gradient convolutions */
  int32_t inference_only;    /* != 0: no buffers for a backward (pre-activations, activation gradients, dz /
                              * weight-gradient scratch, data-gradient weights); mimo_forward then requires
                              * training = 0 and no_grad = 1.  What torch.no_grad() + eval() means for memory. */
  float center_dropout_rate, final_dropout_rate; /* element-wise nn.Dropout (model.py:213, :277-281): the rates the
                              * in-engine generator uses for those sites (mimo_forward_args.rng_sites) */
  int32_t norm_kind, act_kind, up_kind; /* mimo_norm_kind / mimo_act_kind / mimo_up_kind: 0 = the reference's blocks */
} mimo_config;

const char* mimo_last_error(void);
int mimo_version(void);

/* ---- plan life cycle: replaces MimoUNet.__init__ (model.py:31-92) ---------------------- */
int mimo_plan_create(const mimo_config* cfg, mimo_plan** out);
void mimo_plan_destroy(mimo_plan* plan);
size_t mimo_plan_workspace_bytes(const mimo_plan* plan);

/* Parameter inventory in canonical order, with the reference's state_dict names and OIHW
 * shapes (probe of MimoUNet.state_dict(); SURVEY §5 checkpoint row).  kind: 0 = trainable
 * parameter (offset into the flat parameter / gradient buffers), 1 = BatchNorm running
 * buffer (offset into the flat buffer array).  Offsets are in floats. */
int mimo_plan_num_tensors(const mimo_plan* plan);
int mimo_plan_tensor_info(const mimo_plan* plan, int index, char* name, int name_cap, int64_t shape[4],
                          int* ndim, int* kind, int64_t* offset);
int64_t mimo_plan_param_floats(const mimo_plan* plan);
int64_t mimo_plan_buffer_floats(const mimo_plan* plan);

/* Bind the torch-owned flat storage.  grads may be NULL for inference-only plans. */
int mimo_plan_bind(mimo_plan* plan, float* params, float* grads, float* bn_buffers);

/* ---- forward: replaces MimoUNet.forward (model.py:94-117) ------------------------------
 * x is NCHW-strided fp32: element (n,s,c,y,x) at x[n*stride_n + s*stride_s + c*H*W + y*W + x]
 * (stride_s = 0 gives repeat_subnetworks, utils.py:51-61).  perm (int64 [S][N], device) is
 * the gather of apply_input_transform (utils.py:38-41) fused into the load; NULL = identity.
 * out is [N,S,Co,H,W] contiguous.  training: BatchNorm uses batch statistics and updates
 * the running buffers.  drop_masks: host array with one device pointer per DoubleConv (forward
 * order, see mimo_plan_num_double_convs), each [N][Cout] multipliers (0 or 1/(1-p)) or NULL for
 * "no dropout at this site" — Dropout2d (components.py:29) incl. MC-dropout (ensemble.py:54-66). */
typedef struct mimo_forward_args {
  const float* x;
  int64_t stride_n, stride_s;
  const int64_t* perm;
  int32_t training;
  const float* const* drop_masks; /* host array [num_double_convs] or NULL */
  float* out;
  /* element-wise nn.Dropout multipliers (0 or 1/(1-p)), reference layout, or NULL (= none):
   * host array [1 + S] of device pointers, each NULL for "not active":
   *   [0]     center_dropout (model.py:213)        [N][8fS][H/16][W/16]  on down4's output
   *   [1 + s] final_dropouts[s] (model.py:277-281) [N][f][H][W]          in front of head s   */
  const float* const* elem_masks;
  /* Inference fast path (no reference counterpart; what `torch.no_grad()` + `model.eval()` buys the
   * reference is only skipped autograd bookkeeping):
   *   no_grad != 0 with training == 0: no backward will follow this forward — eval-mode BatchNorm +
   *     ReLU (+ Dropout2d multipliers) are folded into the convolution epilogue and the pre-activation
   *     tensors are not stored; mimo_backward after such a forward fails with MIMO_ERR_STATE.
   *   param_version: any value that changes whenever the bound parameters or BatchNorm buffers may have
   *     changed (0 = unknown: re-derive everything).  While it stays the same, eval-mode forwards reuse
   *     the packed weight copies and BatchNorm scale/shift of the previous call. */
  int32_t no_grad;
  int64_t param_version;
  /* In-engine dropout (replaces the Bernoulli draws inside nn.Dropout2d / nn.Dropout, components.py:29,
   * model.py:213,277-281, incl. the MC-dropout passes of ensemble.py:54-66).  rng_sites: host array
   * [num_double_convs + 1 + S] or NULL; a non-zero entry makes the engine draw that site's multipliers itself
   * (Dropout2d sites in DoubleConv order with the config's encoder / core / decoder rates, then center_dropout, then
   * final_dropouts[s]) from a Philox4x32-10 stream keyed by (rng_seed, rng_offset) — take both from the caller's
   * generator and advance its offset.  The backward regenerates the element-wise multipliers from the same pair.
   * Sites given through drop_masks / elem_masks keep their recorded multipliers (parity tests). */
  const uint8_t* rng_sites;
  uint64_t rng_seed, rng_offset;
  /* batch rows of the x tensor (= of the label / mask tensors later given to mimo_loss_forward); 0 = the plan's
   * batch.  With perm the gather may repeat rows (batch_repetitions, utils.py:27-31), so x can hold fewer rows than
   * the plan's batch; the staged (hipGraph) paths copy exactly this many. */
  int64_t x_rows;
} mimo_forward_args;
/* The dropout multipliers of one site as the last mimo_forward used them (tests, and recording a run for bit-level
 * replay through drop_masks / elem_masks): site < num_double_convs -> [N][Cout]; num_double_convs + j -> the
 * element-wise site j (0 = center, 1 + s = final s) in the reference's layout [N][C][H'][W']. */
int mimo_plan_dropout_mask(mimo_plan* plan, int site, float* dst, mimo_stream stream);

/* Numerics status of the plan's kernels since the last clear — the reference has no counterpart (torch raises
 * nothing either; a diverged or out-of-range run shows as NaN losses): the default split16 forward carries
 * activations as fp16 (hi, lo) pairs and weights as fp16 pairs x 2^8, so |activation| >= 65520 or |weight| >= 256
 * overflows where the reference's fp32 path (mimo/models/mimo_components/components.py:23-29) does not.
 * Bits of *flags: MIMO_STATUS_FWD_STATS — a BatchNorm batch statistic of a training forward was not finite;
 * MIMO_STATUS_BWD_STATS — a BatchNorm-backward sum was not finite; MIMO_STATUS_LOGITS — a logit was not finite.
 * The kernels record them for free (one test per finalized channel sum / logit); this call copies the word to the
 * host and is the one plan call that synchronises `stream`.  clear != 0 resets the word. */
#define MIMO_STATUS_FWD_STATS 1
#define MIMO_STATUS_BWD_STATS 2
#define MIMO_STATUS_LOGITS 4
int mimo_plan_status(mimo_plan* plan, int32_t* flags, int32_t clear, mimo_stream stream);
int mimo_plan_num_double_convs(const mimo_plan* plan);
int mimo_plan_double_conv_channels(const mimo_plan* plan, int index); /* Cout of DoubleConv #index */
int mimo_forward(mimo_plan* plan, const mimo_forward_args* args, mimo_stream stream);

/* ---- loss: replaces UncertaintyLoss.forward(reduce_mean=False).mean((0,2,3,4))
 * (losses.py:132-164, mimo_unet.py:241-242) on the logits of the last mimo_forward.
 * label [N,Ct,H,W], mask [N,1,H,W] or NULL, both gathered through perm like x.
 * loss_out: device [S]. */
int mimo_loss_forward(mimo_plan* plan, const float* label, const float* mask, const int64_t* perm,
                      float* loss_out, mimo_stream stream);

/* ---- backward: replaces autograd through MimoUNet (+ the loss when dloss != NULL) --------
 * dout  [N,S,Co,H,W] upstream gradient of the logits, or NULL.
 * dloss device [S]: upstream gradient of the per-subnetwork losses returned by
 *       mimo_loss_forward (e.g. weights/S, mimo_unet.py:138), or NULL.  The closed-form
 *       Laplace/Gaussian gradient is evaluated inside the head kernel.
 * dx    [N,S,Ci,H,W] gradient w.r.t. the input (FGSM, scripts/test/test_nyuv2_depth.py:41-55)
 *       or NULL to skip the first-layer dgrad.  Requires perm == NULL in the forward.
 * Parameter gradients are written (not accumulated) into the bound flat grads buffer. */
int mimo_backward(mimo_plan* plan, const float* dout, const float* dloss, float* dx, mimo_stream stream);

/* The same backward in stages, for data-parallel training (no reference counterpart: the reference is
 * single-GPU, scripts/train/train_ndvi.py:70 `devices=1`).  Stages run in order 0 .. n-1:
 *   0 heads + S decoders, 1 up3, 2 up2, 3 up1, 4 down4, 5 down3, 6 down2, 7 the S encoders (+ dx).
 * The parameters of a stage are one contiguous range [begin, end) of the flat gradient buffer (laid out encoder |
 * core | decoder | heads) that is final when the stage returns, so its all-reduce can run while the later stages
 * back-propagate: one bucket per core block, as SURVEY 8(e) asks.  The exchange itself is torch.distributed /
 * RCCL on the caller's side (mimo_unet_amd/ddp.py), not part of this ABI: there is no mimo_ddp_* entry point. */
int mimo_plan_num_backward_stages(const mimo_plan* plan);
int mimo_plan_backward_stage_range(const mimo_plan* plan, int stage, int64_t* begin, int64_t* end);
int mimo_backward_stage(mimo_plan* plan, int stage, const float* dout, const float* dloss, float* dx,
                        mimo_stream stream);
/* The same stage without making `stream` wait for the plan's side stream (the weight gradients run there): the stage's
 * range is final on *ready_stream — the side stream, which has been made to wait for `stream`, or `stream` itself when
 * the plan has no side stream / replays a per-stage graph — and the caller issues that range's collective THERE (under
 * torch: `with torch.cuda.stream(ExternalStream(ready))`), so the main stream goes straight on with the next stage.  The
 * last stage joins the two streams.  (What Lightning DDP's bucket hooks give the reference's module for free on one
 * stream; here the per-stage join of mimo_backward_stage cost 0.45 ms of a 4.4 ms step at 4 images per GPU.) */
int mimo_backward_stage_async(mimo_plan* plan, int stage, const float* dout, const float* dloss, float* dx,
                              mimo_stream stream, mimo_stream* ready_stream);
int64_t mimo_plan_encoder_param_floats(const mimo_plan* plan);

/* ---- measurement (no reference counterpart): per-kernel-class device time from HIP events
 * recorded on the launch stream around every 3x3 convolution launch, with the ALGORITHMIC
 * flops (2*9*Cin*Cout*N*H*W, logical channels) and bytes ((Cin+Cout)*N*H*W*4) of those
 * launches.  mimo_plan_profile(plan, 1) resets and arms; reads synchronise on the events. */
enum mimo_prof_kind {
  MIMO_PROF_CONV_FWD = 0,
  MIMO_PROF_CONV_DGRAD = 1,
  MIMO_PROF_CONV_WGRAD = 2,
  MIMO_PROF_BN_RELU_FWD = 3,   /* bandwidth class: BatchNorm + ReLU forward pass (8 B per element) */
  MIMO_PROF_BN_BWD_REDUCE = 4, /* BatchNorm backward pass 1 (8 B per element) */
  MIMO_PROF_BN_BWD_APPLY = 5,  /* BatchNorm backward pass 2 (12 B per element) */
  MIMO_PROF_UPCAT_FWD = 6,     /* bilinear x2 + pad + concat (writes the up-sampled channels) */
  MIMO_PROF_UP_BWD = 7,        /* its gradient gather */
  MIMO_PROF_POOL_BWD = 8,      /* MaxPool2d backward (+ skip-gradient fold) */
  MIMO_PROF_HEAD_FWD = 9,      /* 1x1 head */
  MIMO_PROF_HEAD_BWD = 10,     /* NLL gradient + 1x1 head backward */
  MIMO_PROF_KINDS = 11
};
int mimo_plan_profile(mimo_plan* plan, int enable);
int mimo_plan_profile_read(mimo_plan* plan, int kind, double* total_ms, int64_t* launches, double* flops,
                           double* bytes);
/* Device time of everything launched for the DoubleConv blocks (+ heads) of one resolution tier
 * (0 = full resolution H x W ... 4 = H/16 x W/16), forward and backward separately, since the profile was
 * armed: the denominator of the per-tier HBM fraction of SURVEY 8(d). */
int mimo_plan_profile_read_tier(mimo_plan* plan, int tier, double* forward_ms, double* backward_ms);
/* device time of one kernel class inside one resolution tier (the per-tier kernel table of bench.py) */
int mimo_plan_profile_read_kind_tier(mimo_plan* plan, int kind, int tier, double* ms);

/* ---- optimiser: replaces torch.optim.Adam.step (mimo_unet.py:186-190; L2-in-grad) ------ */
int mimo_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                   float grad_scale, mimo_stream stream);

/* The same step under a loss scaler (the reference trains with Lightning precision="16-mixed" = torch.cuda.amp.GradScaler,
 * scripts/train/train_ndvi.py:71).  step_dev: device float, the optimiser's step count, advanced here unless
 * *found_inf != 0; amp_scale / found_inf: device floats as GradScaler hands them to a fused optimiser (gradients are
 * divided by *amp_scale; the whole update is skipped when *found_inf != 0) or NULL.  No host synchronisation. */
int mimo_adam_step_amp(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                       float beta1, float beta2, float eps, float weight_decay, float* step_dev, float grad_scale,
                       const float* amp_scale, const float* found_inf, mimo_stream stream);

/* ---- uncertainties: replaces compute_uncertainties (utils.py:76-101) --------------------
 * p1, p2 [N,S,C,HW] contiguous -> mean, aleatoric_var, epistemic_var [N,C,HW]. */
int mimo_uncertainties(const float* p1, const float* p2, int32_t n, int32_t s, int32_t c, int64_t hw,
                       int32_t loss_kind, float* mean, float* aleatoric, float* epistemic, mimo_stream stream);

/* ---- evidential regression head + loss: replaces the tail of EvidentialUnetModel.forward (mimo/models/
 * evidential_unet.py:90-96: mu, softplus(logv), softplus(logalpha) + 1, softplus(logbeta)) and EvidentialLoss.forward
 * (mimo/losses.py:202-247, coeff-free form the reference evaluates) and their autograd backward.
 * logits [N,4,HW] (the S = 1 backbone output), label [N,HW] or NULL, mask [N,HW] or NULL.
 * forward : ev [N,4,HW] = (gamma, v, alpha, beta); loss_map [N,HW] = per-pixel loss (x mask), or NULL.
 * backward: dlogits [N,4,HW] = d_loss [N,HW] (NULL = 0) through the loss  +  d_ev [N,4,HW] (NULL = 0) through the
 *           softplus heads. */
int mimo_evidential_forward(const float* logits, const float* label, const float* mask, int32_t n, int64_t hw, float* ev,
                            float* loss_map, mimo_stream stream);
int mimo_evidential_backward(const float* logits, const float* label, const float* mask, const float* d_ev,
                             const float* d_loss, int32_t n, int64_t hw, float* dlogits, mimo_stream stream);

/* ---- validation epilogue: replaces, after the forward, the tail of MimoUnetModel.validation_step
 * (mimo_unet.py:153-183): compute_uncertainties, sqrt of the variances, calculate_dist_param(log=True) +
 * the combined NLL on the ensemble mean, the error map, compute_regression_metrics (metrics.py:22-34:
 * r2 / mae / mse / rmse) and the two logged uncertainty means — one pass over the logits.
 * out [N,S,2*Ct,HW] logits, label [N,Ct,HW], mask [N,1,HW] or NULL.
 * mean / aleatoric_std / epistemic_std / err: [N,Ct,HW] each.
 * scalars: device [8] = combined NLL, mae, mse, rmse, r2, mean clip(aleatoric_std,0,5),
 *          mean clip(epistemic_std,0,5), element count.
 * scratch: device doubles [scratch_blocks * 8], scratch_blocks >= 1 (1024 is enough for any size). */
int mimo_validation_epilogue(const float* out, const float* label, const float* mask, int32_t n, int32_t s, int32_t ct,
                             int64_t hw, int32_t loss_kind, float eps_min, float eps_max, float* mean,
                             float* aleatoric_std, float* epistemic_std, float* err, float* scalars, double* scratch,
                             int32_t scratch_blocks, mimo_stream stream);

/* ---- loss-buffer step: replaces MimoUnetModel._calculate_train_loss's arithmetic on the [S] loss vector
 * (mimo_unet.py:223-247: weights = loss_buffer.get_weights() read BEFORE loss_buffer.add(loss); loss_buffer.py:43-74:
 * ring mean over ALL rows incl. still-zero ones, softmax(mean / T) * S) — one launch instead of ~9 tensor operations.
 * ring    device [size][S], updated in place: row `index` <- loss (the caller advances its index, loss_buffer.py:52)
 * loss    device [S] (mimo_loss_forward's output); weights / w_over_s: device [S] each = the weights, and weights / S
 *         (the gradient of mean(loss * weights) w.r.t. loss: mimo_backward's dloss); scalars: device [2] =
 *         mean(loss * weights), mean(loss).  1 <= S <= 64. */
int mimo_loss_buffer_step(float* ring, int32_t size, int32_t index, int32_t s, float temperature, const float* loss,
                          float* weights, float* w_over_s, float* scalars, mimo_stream stream);

/* ---- training-step epilogue: replaces, after the fused forward + loss, the no_grad tail of
 * MimoUnetModel.training_step (mimo_unet.py:121-144): the per-subnetwork label gather of apply_input_transform
 * (utils.py:38-48), loss_fn.mode / loss_fn.std (losses.py:166-192), the error map and compute_regression_metrics
 * on the flattened predictions (metrics.py:22-34) — one pass instead of ~20 tensor operations.
 * out [N,S,2*Ct,HW] logits; label [N0,Ct,HW]; perm [S][N] int64 rows of label per (s, n), or NULL (identity, N0 = N).
 * label_t / preds / aleatoric_std / err: [N,S,Ct,HW] each.
 * scalars: device [5] = mae, mse, rmse, r2, element count.  scratch: device doubles [scratch_blocks * 8]. */
int mimo_training_epilogue(const float* out, const float* label, const int64_t* perm, int32_t n, int32_t s, int32_t ct,
                           int64_t hw, int32_t loss_kind, float* label_t, float* preds, float* aleatoric_std, float* err,
                           float* scalars, double* scratch, int32_t scratch_blocks, mimo_stream stream);

/* ---- single-operator entry points (NHWC, channel-padded) used by the parity tests -------
 * They run the same kernels the plan runs.  x [N,H,W,cin_p], w OIHW [cout][cin][3][3]. */
int mimo_op_conv3x3_forward(const float* x, const float* w, const float* bias, float* z, double* stats,
                            int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cin_p, int32_t cout,
                            int32_t cout_p, int32_t precision, mimo_stream stream);
int mimo_op_conv3x3_dgrad(const float* dz, const float* w, float* dx, int32_t n, int32_t h, int32_t wd,
                          int32_t cin, int32_t cin_p, int32_t cout, int32_t cout_p, int32_t precision,
                          mimo_stream stream);
int mimo_op_conv3x3_wgrad(const float* x, const float* dz, float* dw, float* dbias, int32_t n, int32_t h,
                          int32_t wd, int32_t cin, int32_t cin_p, int32_t cout, int32_t cout_p, int32_t precision,
                          mimo_stream stream);
int mimo_op_maxpool2x2(const float* x, float* y, int32_t n, int32_t h, int32_t w, int32_t c_p, mimo_stream stream);
int mimo_op_upsample_cat(const float* skip, const float* low, float* out, int32_t n, int32_t hs, int32_t ws,
                         int32_t cs_p, int32_t hl, int32_t wl, int32_t cl_p, mimo_stream stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* MIMO_HIP_H */
