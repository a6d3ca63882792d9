// Launch wrappers of the bandwidth-class kernels (elementwise.hip).  All tensors NHWC, channel count padded to a
// multiple of 8; "ld" = ELEMENTS per pixel of the buffer a pointer indexes (a tensor may be a channel slice of a
// wider concat buffer).  Activation-like tensors (conv outputs z, activations a, their gradients da / dxpad / dz)
// are fp32, or — in the 16-bit storage modes (mimo_precision *_MIXED) — bf16 / fp16: such pointers are passed
// untyped together with a StoreType ("dt"; "dtz" for a conv output, which stays fp32 for the image convolution).
// Statistics, masks, BatchNorm vectors, logits and every accumulation stay fp32.
#pragma once
#include "common.h"

namespace mimo {

constexpr int kMaxChunks = 64;   // second-level partial rows kept for finalize kernels
constexpr int kEwMaxBlocks = 1024;
constexpr int kBnReduceMaxBlocks = 2048;  // upper bound of the BatchNorm-backward grids (partial-row capacity; 1792 used)
constexpr int kMaxHeadOut = 8;   // out_channels supported by the fused head kernels

// ---- generic two-level column reduction of per-workgroup partial rows -------------------
// partial [rows][cols] (float) -> sums [chunks][cols] (double); returns chunks via *chunks.
int rowsum_launch(const float* partial, int rows, int cols, double* sums, int* chunks, hipStream_t s);
// one-launch reductions over fp32 partial rows (column sums + the finalize arithmetic); use while rows <= kColsumMaxRows.
// ColsumScratch: doubles [kMaxChunks][2][round_up(columns, 64)] + one zero-initialised int ticket per 64-column group
// (kColsumMaxGroups), owned by the plan; launches sharing it must be ordered on one stream.
constexpr int kColsumMaxRows = 2048;
constexpr int kColsumMaxGroups = 256;
struct ColsumScratch {
  double* sums;
  int* tickets;
  int* status = nullptr;  // numerics status word of the plan (mimo_plan_status): OR-ed when a finalized sum is not finite
};
// bits of the status word
constexpr int kStatusFwdStats = 1, kStatusBwdStats = 2, kStatusLogits = 4;
int colsum_vec_launch(const float* partial, int rows, int cols, int C, float* out, const ColsumScratch& cs, hipStream_t st);
int bn_fwd_stats_launch(const float* partial, int rows, int cout_pad, int C, int Cp, int64_t count, const float* gamma,
                        const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* mean,
                        float* invstd, float* scale, float* shift, const ColsumScratch& cs, hipStream_t st);
int bn_bwd_stats_launch(const float* partial, int rows, int C, int Cp, int64_t count, int training, float* c1, float* c2,
                        float* dgamma, float* dbeta, float* dbias_zero, const ColsumScratch& cs, hipStream_t st);
int head_bwd_stats_launch(const float* partial, int rows, int C, int Cp, int Co, float* dw, float* db, const ColsumScratch& cs,
                          hipStream_t st);

// ---- input / output layout conversion ---------------------------------------------------
// x NCHW-strided (see mimo_forward_args) -> NHWC [N,H,W,cp] for subnetwork s (zero pad channels)
int pack_input_launch(const float* x, int64_t stride_n, int64_t stride_s, const int64_t* perm, int s, int N, int C,
                      int H, int W, float* out, int cp, hipStream_t st);
// dx[n][s][c][y][x] = fold(dxpad)[n][y][x][c]
int unpack_dx_launch(const void* dxpad, int dt, int ldp, int N, int S, int s, int C, int H, int W, float* dx, hipStream_t st);

// ---- BatchNorm + ReLU (+ Dropout2d) forward ---------------------------------------------
// training: sums = rowsum of the conv epilogue partials ([chunks][2*cout_pad]).
int bn_fwd_finalize_launch(const double* sums, int chunks, int cout_pad, int C, int Cp, int64_t count,
                           const float* gamma, const float* beta, float* running_mean, float* running_var,
                           float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                           hipStream_t st);
int bn_eval_prepare_launch(int C, int Cp, const float* gamma, const float* beta, const float* running_mean,
                           const float* running_var, float eps, float* mean, float* invstd, float* scale,
                           float* shift, hipStream_t st, int* status = nullptr);
// a = relu(z*scale+shift) * mask[n][c]
// status != nullptr (eval mode): kStatusFwdStats is OR-ed into it when an input element is not finite (fmaxf would drop it)
int bn_relu_fwd_launch(const void* z, int dtz, int ldz, void* a, int dta, int lda, const float* scale, const float* shift,
                       const float* mask, int C, int Cp, int64_t P, int HW, hipStream_t st, int* status = nullptr);

// same, for a tensor that feeds MaxPool2d(2): also writes pool[N,H/2,W/2] = maxpool2x2(a) (one pass, 2x2 window per thread)
int bn_relu_pool_fwd_launch(const void* z, int dtz, int ldz, void* a, int dta, int lda, const float* scale, const float* shift,
                            const float* mask, int C, int Cp, int N, int H, int W, void* pool, int ldpool, hipStream_t st,
                            int* status = nullptr);

// ---- pooling / upsampling ---------------------------------------------------------------
int maxpool_fwd_launch(const void* a, int dt, int lda, int N, int H, int W, int Cp, void* out, int ldo, hipStream_t st);
// out[N,H,W,csp+clp] = cat(skip, zero_pad(bilinear_x2_align_corners(low))); skip == nullptr: channels [0, csp) of
// out already hold the skip tensor (its producer writes it in place), only the up-sampled part is written
// lo_scale / lo_shift != nullptr: `low` is the producing convolution's pre-activation tensor, its BatchNorm + ReLU is applied here
int upcat_fwd_launch(const void* skip, int dt, int lds, int csp, const void* low, int ldl, int clp, int N, int H, int W,
                     int h, int w, void* out, hipStream_t st, const float* lo_scale = nullptr, const float* lo_shift = nullptr);

// ---- backward gathers.  "dxpad" = gradient on the reflect-padded domain [N,H+2,W+2,ldp]
// produced by the dgrad convolution; fold = transpose of reflect padding. ------------------
// da[N,H,W] (lda) (=|+=) maxpool2x2 backward of fold(dxpad at pooled size), argmax from a
// skip != nullptr: + fold(skip [N,H+2,W+2] (ldsk))[..., 0 : Cp] — the skip-connection gradient of the same tensor,
// read straight from the Up block's padded-domain data gradient
int pool_bwd_launch(const void* dxpad, int dt, int ldp, int choff, const void* a, int lda, void* da, int ldda, int N,
                    int H, int W, int Cp, int accumulate, hipStream_t st, const void* skip = nullptr, int ldsk = 0);
// da (=|+=) fold(dxpad)[..., choff : choff+Cp]
int fold_slice_launch(const void* dxpad, int dt, int ldp, int choff, void* da, int ldda, int N, int H, int W, int Cp,
                      int accumulate, hipStream_t st);
// da_low[N,h,w] (=|+=) bilinear^T(fold(dxpad [N,H+2,W+2])[..., choff : choff+Cp])
int up_bwd_launch(const void* dxpad, int dt, int ldp, int choff, void* da, int ldda, int N, int H, int W, int h, int w,
                  int Cp, int accumulate, hipStream_t st);

// a[N,HW] (ld) *= mask, mask in the reference's NCHW layout [N][C][HW] (nn.Dropout multipliers)
// mask == nullptr: multipliers from the Philox stream `rng` (same values in forward and backward)
struct ElemRng {
  uint64_t seed, offset;
  int site;  // distinguishes the streams of the sites sharing one (seed, offset)
  float p;   // drop probability
};
int elem_mask_mul_launch(void* a, int dt, int ld, const float* mask, int N, int C, int Cp, int HW, hipStream_t st,
                         const ElemRng* rng = nullptr);
// the same multipliers written out in the reference's layout [N][C][HW]
int elem_dropout_mask_launch(float* mask, int N, int C, int Cp, int HW, int site, uint64_t seed, uint64_t offset, float p,
                             hipStream_t st);
// Dropout2d (one Bernoulli per (sample, channel), components.py:29): multipliers of all active sites in one launch
struct Dropout2dSite {
  float* dst;  // [count] = [N][C]
  int count;
  float p;
};
int dropout2d_masks_launch(const Dropout2dSite* sites_dev, int nsites, int max_count, uint64_t active, uint64_t seed,
                           uint64_t offset, hipStream_t st);

// test hook: one idle wave for `us` microseconds on `st` (MIMO_DEBUG_WGRAD_DELAY_US)
int debug_delay_launch(int us, hipStream_t st);

// ---- BatchNorm + ReLU backward ------------------------------------------------------------
// dy = (gradient arriving at the activation) * mask * [z*scale+shift > 0] is evaluated on the fly by both passes (never
// stored).  Where that gradient comes from:
enum GradSrcKind {
  GS_PLAIN = 0,  // a plain NHWC buffer (da, ldda)
  GS_FOLD = 1,   // the padded-domain data gradient of the next convolution, folded (dxpad, ldp)
  // the tensor is pooled by a Down block (and possibly the skip input of an Up block): MaxPool2d backward of
  // fold(dxpad at the pooled size)[choff ...] — the window's activations are re-formed from z, the arithmetic of
  // bn_relu_pool_fwd_kernel — plus fold(skip)[skoff ...]: what pool_bwd_kernel would have written, never written
  // (fp32 storage; components.py:48 backward + the skip half of torch.cat's backward, components.py:118)
  GS_POOL = 2,
  // the tensor feeds the 1x1 head: W^T (dout + dloss/count * dNLL/dlogit) formed per pixel from the logits and labels —
  // what head_bwd_kernel would have written; the reduce pass also accumulates the head's weight / bias gradient
  // partial rows [rows][Co*Cp + Co] (Co == 2, fp32 storage; components.py:126 backward + losses.py:151-160)
  GS_HEAD = 3,
};
struct HeadGrad {
  const float* w;
  int C, Co, N, S, s, HW;
  const float *out, *dout, *dloss, *label, *mask;
  const int64_t* perm;
  int kind;
  float eps_min, eps_max, inv_count;
  float* partial;  // reduce pass: the head's weight / bias gradient partial rows
};
struct GradSrc {
  int kind = GS_PLAIN;
  const void* da = nullptr;
  int ldda = 0;
  const void* dxpad = nullptr;
  int ldp = 0, choff = 0;
  const void* skip = nullptr;
  int ldsk = 0, skoff = 0;
  HeadGrad head = {};
  static GradSrc plain(const void* da, int ldda) {
    GradSrc g;
    g.da = da;
    g.ldda = ldda;
    return g;
  }
  static GradSrc fold(const void* dxpad, int ldp) {
    GradSrc g;
    g.kind = GS_FOLD;
    g.dxpad = dxpad;
    g.ldp = ldp;
    return g;
  }
};
// pass 1: partial rows of (sum dy, sum dy*xhat)
int bnrelu_bwd_reduce_launch(const GradSrc& src, int dta, const void* z, int dtz, int ldz,
                             const float* scale, const float* shift, const float* mean, const float* invstd,
                             const float* mask, int C, int Cp, int N, int H, int W, float* partial, int* rows,
                             hipStream_t st);
// c1 = sum_dy/count, c2 = sum_dyxhat/count (zero when !training); dgamma, dbeta -> grads
int bn_bwd_finalize_launch(const double* sums, int chunks, int C, int Cp, int64_t count, int training, float* c1,
                           float* c2, float* dgamma, float* dbeta, hipStream_t st);
// pass 2: dz = scale * (dy - c1 - xhat*c2); partial rows of sum dz (conv bias gradient)
int bn_bwd_apply_launch(const GradSrc& src, int dta, const void* z, int dtz, int ldz,
                        const float* scale, const float* shift, const float* mean, const float* invstd, const float* mask,
                        int C, const float* c1, const float* c2, int Cp, int N, int H, int W, void* dz, int split_out,
                        float* partial, int* rows, hipStream_t st, float* absmax = nullptr,
                        int* absmax_n = nullptr,  // absmax: *absmax_n per-workgroup maxima of |dz| (WgradLaunch::dz_absmax, <= kDzMaxSlots)
                        hipEvent_t done = nullptr);  // != nullptr: recorded when the kernel completes (attached to the launch)
// dst[p] = [hi Cp bf16 | lo Cp bf16] of src[p][Cp] — the storage the bf16-pair convolution kernels read
// (split_out != 0 above writes dz in this form directly)
// absmax != nullptr: *absmax_n per-workgroup maxima of |src| are left there (WgradLaunch::dz_absmax; capacity kDzMaxSlots floats)
int split_pairs_launch(const float* src, float* dst, int64_t P, int Cp, hipStream_t st, float* absmax = nullptr,
                       int* absmax_n = nullptr);
// out[c] = sum over chunks of sums[chunk][c], c < C
int vec_finalize_launch(const double* sums, int chunks, int cols, int C, float* out, hipStream_t st);

// ---- 1x1 head + loss ------------------------------------------------------------------------
// out[n][s][co][yx] = bias[co] + sum_c a[n,yx,c] * w[co][c]      (components.py:126)
// status != nullptr: kStatusLogits is OR-ed into it when a logit is not finite
// in_scale / in_shift != nullptr (head_fwd / head_bwd): `a` is the pre-activation tensor of the decoder's last convolution and
// its BatchNorm + ReLU is applied on the way in (relu(fma(z, scale, shift)): bn_relu_fwd_kernel's arithmetic)
int head_fwd_launch(const void* a, int dt, int lda, const float* w, const float* bias, int C, int Co, int N, int S, int s,
                    int HW, float* out, hipStream_t st, int* status = nullptr, const float* in_scale = nullptr,
                    const float* in_shift = nullptr);
// per-subnetwork sum of the un-reduced NLL (losses.py:151-160) -> partial [S][blocks]
int loss_fwd_launch(const float* out, const float* label, const float* mask, const int64_t* perm, int N, int S,
                    int Co, int HW, int kind, float eps_min, float eps_max, float* partial, int* blocks,
                    hipStream_t st);
int loss_finalize_launch(const float* partial, int S, int blocks, double count, float* loss_out, hipStream_t st);
// dlogit = dout + dloss[s]/count * dNLL/dlogit; da = W^T dlogit; partial rows of (dW [Co][Cp], db [Co])
int head_bwd_launch(const void* a, int dt, int lda, const float* w, int C, int Cp, int Co, int N, int S, int s, int HW,
                    const float* out, const float* dout, const float* dloss, const float* label, const float* mask,
                    const int64_t* perm, int kind, float eps_min, float eps_max, void* da, float* partial,
                    int* rows, hipStream_t st, const float* in_scale = nullptr, const float* in_shift = nullptr);
int head_bwd_finalize_launch(const double* sums, int chunks, int C, int Cp, int Co, float* dw, float* db,
                             hipStream_t st);

}  // namespace mimo
