// Fused flat-buffer Adam and the subnetwork-axis uncertainty reduction (gfx950, HBM-bound).
//
// Replaces: torch.optim.Adam.step as configured by mimo/models/mimo_unet.py:186-190
//           (betas .9/.999, eps 1e-8, L2 weight decay folded into the gradient, not AdamW);
//           compute_uncertainties, mimo/models/utils.py:76-101.
#include <algorithm>
#include <cmath>

#include "common.h"

namespace mimo {

// amp (loss-scaled training, torch.cuda.amp.GradScaler protocol): step_dev counts the optimiser steps ON THE DEVICE —
// amp_step_kernel advances it unless the scaler found an inf / nan — and the update below then derives its bias
// corrections from it, divides the gradients by *amp_scale and is skipped as a whole when *found_inf != 0: no
// host synchronisation anywhere in the step.
__global__ void amp_step_kernel(float* __restrict__ step_dev, const float* __restrict__ found_inf) {
  if (!found_inf || *found_inf == 0.f) *step_dev += 1.f;
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float lr, float beta1, float beta2, float eps, float wd,
                            float bc1, float bc2_sqrt, float grad_scale, const float* __restrict__ step_dev,
                            const float* __restrict__ amp_scale, const float* __restrict__ found_inf) {
  if (found_inf && *found_inf != 0.f) return;  // the scaler saw an inf / nan gradient: skip this step
  if (step_dev) {
    const float t = *step_dev;
    bc1 = 1.f - powf(beta1, t);
    bc2_sqrt = sqrtf(1.f - powf(beta2, t));
  }
  if (amp_scale) grad_scale /= *amp_scale;
  const int64_t n4 = n / 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* pa = &pp.x;
    float* ga = &gg.x;
    float* ma = &mm.x;
    float* va = &vv.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float gr = ga[j] * grad_scale;
      if (wd != 0.f) gr = fmaf(wd, pa[j], gr);
      ma[j] = beta1 * ma[j] + (1.f - beta1) * gr;
      va[j] = beta2 * va[j] + (1.f - beta2) * gr * gr;
      const float denom = sqrtf(va[j]) / bc2_sqrt + eps;
      pa[j] -= (lr / bc1) * (ma[j] / denom);
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  // tail (n not a multiple of 4)
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float gr = g[i] * grad_scale;
    if (wd != 0.f) gr = fmaf(wd, p[i], gr);
    const float mi = beta1 * m[i] + (1.f - beta1) * gr;
    const float vi = beta2 * v[i] + (1.f - beta2) * gr * gr;
    m[i] = mi;
    v[i] = vi;
    p[i] -= (lr / bc1) * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  }
}

__global__ void uncertainty_kernel(const float* __restrict__ p1, const float* __restrict__ p2, int N, int S, int64_t chw,
                                   int kind, float* __restrict__ mean, float* __restrict__ alea, float* __restrict__ epi) {
  const int64_t total = (int64_t)N * chw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / chw, r = i - n * chw;
    const float* a = p1 + n * S * chw + r;
    const float* b = p2 + n * S * chw + r;
    float sm = 0.f, sa = 0.f;
    for (int s = 0; s < S; ++s) {
      sm += a[s * chw];
      // std^2: Laplace (exp(ls)*sqrt2)^2, Gaussian (exp(lv)^0.5)^2  (losses.py:82-84,166-167)
      const float e = expf(b[s * chw]);
      const float sd = kind == MIMO_LOSS_LAPLACE_NLL ? e * 1.41421356237309515f : sqrtf(e);
      sa += sd * sd;
    }
    const float mu = sm / (float)S;
    float se = 0.f;
    for (int s = 0; s < S; ++s) {
      const float d = a[s * chw] - mu;
      se += d * d;
    }
    mean[i] = mu;
    alea[i] = sa / (float)S;
    epi[i] = S > 1 ? se / (float)(S - 1) : 0.f;
  }
}

// ---------------------------------------------------------------------------------------
// Validation epilogue (mimo_unet.py:146-183 after the forward, metrics.py:22-34): one pass over the
// logits produces the ensemble mean, the aleatoric / epistemic standard deviations, the error map and
// the sums behind the combined NLL, mae / mse / rmse / r2 and the logged uncertainty means; a
// one-workgroup second pass turns the per-workgroup partials (double) into the eight scalars.
// ---------------------------------------------------------------------------------------
constexpr int kValSums = 8;  // nll, err^2, |err|, y, y^2, clip(alea_std), clip(epi_std), (unused)

__global__ void val_epilogue_kernel(const float* __restrict__ out, const float* __restrict__ label,
                                    const float* __restrict__ mask, int N, int S, int Ct, int64_t hw, int kind, float eps_min,
                                    float eps_max, float* __restrict__ mean, float* __restrict__ alea_std,
                                    float* __restrict__ epi_std, float* __restrict__ err, double* __restrict__ partial) {
  __shared__ double red[kValSums][256];
  const int64_t chw = (int64_t)Ct * hw, total = (int64_t)N * chw;
  double acc[kValSums];
#pragma unroll
  for (int k = 0; k < kValSums; ++k) acc[k] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / chw, r = i - n * chw;  // r = c * hw + pixel
    const float* a = out + n * S * 2 * chw + r;   // p1[n][s][c] at stride 2*chw per subnetwork
    const float* b = a + chw;                     // p2 = channels Ct .. 2Ct-1 of the same subnetwork
    float sm = 0.f, sa = 0.f;
    for (int s = 0; s < S; ++s) {
      sm += a[(int64_t)s * 2 * chw];
      const float e = expf(b[(int64_t)s * 2 * chw]);
      const float sd = kind == MIMO_LOSS_LAPLACE_NLL ? e * 1.41421356237309515f : sqrtf(e);
      sa += sd * sd;
    }
    const float mu = sm / (float)S;
    float se = 0.f;
    for (int s = 0; s < S; ++s) {
      const float d = a[(int64_t)s * 2 * chw] - mu;
      se += d * d;
    }
    const float av = sa / (float)S, ev = S > 1 ? se / (float)(S - 1) : 0.f;
    const float as = sqrtf(av), es = sqrtf(ev), cs = sqrtf(av + ev);
    const float y = label[i];
    const float d = mu - y;
    // combined NLL: calculate_dist_param(std, log=True) then forward() clamps exp() of it again
    float prm = kind == MIMO_LOSS_LAPLACE_NLL ? cs / 1.41421356237309515f : cs * cs;
    prm = fminf(fmaxf(prm, eps_min), eps_max);
    const float sc = fminf(fmaxf(expf(logf(prm)), eps_min), eps_max);
    float nll = logf(sc) + (kind == MIMO_LOSS_LAPLACE_NLL ? fabsf(d) : d * d) / sc;
    if (mask) nll *= mask[n * hw + (r % hw)];
    mean[i] = mu;
    alea_std[i] = as;
    epi_std[i] = es;
    err[i] = d;
    acc[0] += nll;
    acc[1] += (double)d * d;
    acc[2] += fabsf(d);
    acc[3] += y;
    acc[4] += (double)y * y;
    acc[5] += fminf(fmaxf(as, 0.f), 5.f);
    acc[6] += fminf(fmaxf(es, 0.f), 5.f);
  }
#pragma unroll
  for (int k = 0; k < kValSums; ++k) red[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off)
#pragma unroll
      for (int k = 0; k < kValSums; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x < kValSums) partial[(size_t)blockIdx.x * kValSums + threadIdx.x] = red[threadIdx.x][0];
}

// scalars: [0] combined NLL (mean), [1] mae, [2] mse, [3] rmse, [4] r2, [5] mean clip(aleatoric_std, 0, 5),
//          [6] mean clip(epistemic_std, 0, 5), [7] element count
__global__ void val_finalize_kernel(const double* __restrict__ partial, int blocks, double count, float* __restrict__ scalars) {
  __shared__ double red[kValSums][256];
  __shared__ double tot[kValSums];
  double a[kValSums];
#pragma unroll
  for (int k = 0; k < kValSums; ++k) a[k] = 0.0;
  for (int b = threadIdx.x; b < blocks; b += 256)
#pragma unroll
    for (int k = 0; k < kValSums; ++k) a[k] += partial[(size_t)b * kValSums + k];
#pragma unroll
  for (int k = 0; k < kValSums; ++k) red[k][threadIdx.x] = a[k];
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off)
#pragma unroll
      for (int k = 0; k < kValSums; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x < kValSums) tot[threadIdx.x] = red[threadIdx.x][0];
  __syncthreads();
  if (threadIdx.x == 0) {
    const double mse = tot[1] / count;
    const double ss_tot = tot[4] - tot[3] * tot[3] / count;
    scalars[0] = (float)(tot[0] / count);
    scalars[1] = (float)(tot[2] / count);
    scalars[2] = (float)mse;
    scalars[3] = (float)sqrt(mse);
    scalars[4] = (float)(1.0 - tot[1] / ss_tot);
    scalars[5] = (float)(tot[5] / count);
    scalars[6] = (float)(tot[6] / count);
    scalars[7] = (float)count;
  }
}

// Training-step epilogue (mimo_unet.py:121-144): gathered labels, predictions, aleatoric std, error map and the
// sums behind mae / mse / rmse / r2 over all N*S*Ct*HW elements.
__global__ void train_epilogue_kernel(const float* __restrict__ out, const float* __restrict__ label,
                                      const int64_t* __restrict__ perm, int N, int S, int Ct, int64_t hw, int kind,
                                      float* __restrict__ label_t, float* __restrict__ preds, float* __restrict__ alea_std,
                                      float* __restrict__ err, double* __restrict__ partial) {
  __shared__ double red[4][256];
  const int64_t chw = (int64_t)Ct * hw, total = (int64_t)N * S * chw;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t ns = i / chw, r = i - ns * chw;  // r = c * hw + pixel
    const int64_t n = ns / S, s = ns - n * S;
    const float mu = out[ns * 2 * chw + r], lp = out[ns * 2 * chw + chw + r];
    const int64_t src = perm ? perm[s * N + n] : n;
    const float y = label[src * chw + r];
    const float e = expf(lp);
    const float d = mu - y;
    label_t[i] = y;
    preds[i] = mu;
    alea_std[i] = kind == MIMO_LOSS_LAPLACE_NLL ? e * 1.41421356237309515f : sqrtf(e);
    err[i] = d;
    acc[0] += fabsf(d);
    acc[1] += (double)d * d;
    acc[2] += y;
    acc[3] += (double)y * y;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off)
#pragma unroll
      for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x < 4) partial[(size_t)blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}

// scalars: [0] mae, [1] mse, [2] rmse, [3] r2, [4] element count
__global__ void train_finalize_kernel(const double* __restrict__ partial, int blocks, double count, float* __restrict__ scalars) {
  __shared__ double red[4][256];
  __shared__ double tot[4];
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < blocks; b += 256)  // 256 threads walk the partial rows, then a fixed-order tree
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] += partial[(size_t)b * 4 + k];
#pragma unroll
  for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = a[k];
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off)
#pragma unroll
      for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x < 4) tot[threadIdx.x] = red[threadIdx.x][0];
  __syncthreads();
  if (threadIdx.x == 0) {
    const double mse = tot[1] / count;
    const double ss_tot = tot[3] - tot[2] * tot[2] / count;
    scalars[0] = (float)(tot[0] / count);
    scalars[1] = (float)mse;
    scalars[2] = (float)sqrt(mse);
    scalars[3] = (float)(1.0 - tot[1] / ss_tot);
    scalars[4] = (float)count;
  }
}

// Loss-buffer step of MimoUnetModel._calculate_train_loss (mimo_unet.py:223-247 over loss_buffer.py:43-74) in one launch:
// weights = softmax(mean over ALL ring rows / T) * S, read BEFORE the current loss is written into row `index`; then
// out = {weights [S], weights / S [S] (the gradient of the weighted mean w.r.t. the losses), mean(loss * weights), mean(loss)}.
// One wave; S <= 64 (one subnetwork per lane).
__global__ void loss_buffer_step_kernel(float* __restrict__ ring, int size, int index, int S, float T,
                                        const float* __restrict__ loss, float* __restrict__ weights,
                                        float* __restrict__ w_over_s, float* __restrict__ scalars) {
  const int s = threadIdx.x;
  const bool on = s < S;
  float m = 0.f;
  if (on) {
    for (int r = 0; r < size; ++r) m += ring[(size_t)r * S + s];
    m = m / (float)size / T;  // (torch: buffer.mean(dim=0) / temperature)
  }
  float mx = on ? m : -INFINITY;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
  const float e = on ? expf(m - mx) : 0.f;
  float den = e;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) den += __shfl_xor(den, d);
  const float w = e / den * (float)S;
  const float l = on ? loss[s] : 0.f;
  float lw = l * w, ls = l;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    lw += __shfl_xor(lw, d);
    ls += __shfl_xor(ls, d);
  }
  if (on) {
    weights[s] = w;
    w_over_s[s] = w / (float)S;
    ring[(size_t)index * S + s] = l;
  }
  if (s == 0) {
    scalars[0] = lw / (float)S;
    scalars[1] = ls / (float)S;
  }
}

}  // namespace mimo

extern "C" int mimo_training_epilogue(const float* out, const float* label, const int64_t* perm, int32_t n, int32_t s,
                                      int32_t ct, int64_t hw, int32_t loss_kind, float* label_t, float* preds,
                                      float* aleatoric_std, float* err, float* scalars, double* scratch,
                                      int32_t scratch_blocks, mimo_stream stream) {
  using namespace mimo;
  if (!out || !label || !label_t || !preds || !aleatoric_std || !err || !scalars || !scratch || n < 1 || s < 1 || ct < 1 ||
      hw < 1 || scratch_blocks < 1) {
    set_error("mimo_training_epilogue: invalid argument");
    return MIMO_ERR_INVALID;
  }
  const int64_t total = (int64_t)n * s * ct * hw;
  const int blocks = (int)std::min<int64_t>(std::min<int64_t>(ceil_div64(total, 256), 2048), scratch_blocks);
  hipLaunchKernelGGL(train_epilogue_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, label, perm, n, s, ct, hw,
                     loss_kind, label_t, preds, aleatoric_std, err, scratch);
  MIMO_KERNEL_CHECK();
  hipLaunchKernelGGL(train_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, blocks, (double)total, scalars);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_validation_epilogue(const float* out, const float* label, const float* mask, int32_t n, int32_t s,
                                        int32_t ct, int64_t hw, int32_t loss_kind, float eps_min, float eps_max, float* mean,
                                        float* aleatoric_std, float* epistemic_std, float* err, float* scalars,
                                        double* scratch, int32_t scratch_blocks, mimo_stream stream) {
  using namespace mimo;
  if (!out || !label || !mean || !aleatoric_std || !epistemic_std || !err || !scalars || !scratch || n < 1 || s < 1 ||
      ct < 1 || hw < 1 || scratch_blocks < 1) {
    set_error("mimo_validation_epilogue: invalid argument");
    return MIMO_ERR_INVALID;
  }
  const int64_t total = (int64_t)n * ct * hw;
  const int blocks = (int)std::min<int64_t>(std::min<int64_t>(ceil_div64(total, 256), 1024), scratch_blocks);
  hipLaunchKernelGGL(val_epilogue_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, label, mask, n, s, ct, hw,
                     loss_kind, eps_min, eps_max, mean, aleatoric_std, epistemic_std, err, scratch);
  MIMO_KERNEL_CHECK();
  hipLaunchKernelGGL(val_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, blocks, (double)total, scalars);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// ---------------------------------------------------------------------------------------
// Evidential regression head + loss (mimo/models/evidential_unet.py:74-96, mimo/losses.py:202-247):
//   (mu, v, alpha, beta) = (l0, softplus(l1), softplus(l2) + 1, softplus(l3))
//   loss = G(alpha) / (v sqrt(beta)) * (2 beta (1 + v) + (2 alpha - 1) v (y - mu)^2) + (y - mu)^2 (2 alpha + v),
//   G(alpha) = Gamma(alpha - 1/2) / (4 Gamma(alpha))
// One pass for the four NIG parameters + the per-pixel loss, one pass for the gradient w.r.t. the logits (analytic:
// dG/dalpha = G (psi(alpha - 1/2) - psi(alpha))) instead of ~40 element-wise tensor operations and their autograd
// nodes.  G is evaluated as exp(lgamma(alpha - 1/2) - lgamma(alpha)) / 4 — the reference exponentiates the two
// lgammas separately, which overflows fp32 (inf / inf = nan) for alpha > 35; this form stays finite there.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }  // torch: threshold 20
__device__ __forceinline__ float sigmoid_f(float x) { return x > 20.f ? 1.f : 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float digamma_f(float x) {  // x > 0.5 here (alpha > 1)
  float r = 0.f;
  while (x < 6.f) {
    r -= 1.f / x;
    x += 1.f;
  }
  const float i = 1.f / x, i2 = i * i;
  return r + logf(x) - 0.5f * i - i2 * (1.f / 12.f - i2 * (1.f / 120.f - i2 * (1.f / 252.f)));
}

struct NigPoint {
  float mu, v, alpha, beta, c, T, d;  // c = G / (v sqrt(beta)), T = the bracket, d = y - mu
};
__device__ __forceinline__ NigPoint nig_point(float l0, float l1, float l2, float l3, float y) {
  NigPoint q;
  q.mu = l0;
  q.v = softplus_f(l1);
  q.alpha = softplus_f(l2) + 1.f;
  q.beta = softplus_f(l3);
  q.d = y - q.mu;
  const float G = 0.25f * expf(lgammaf(q.alpha - 0.5f) - lgammaf(q.alpha));
  q.c = G / (q.v * sqrtf(q.beta));
  q.T = 2.f * q.beta * (1.f + q.v) + (2.f * q.alpha - 1.f) * q.v * q.d * q.d;
  return q;
}

__global__ void evidential_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ label,
                                      const float* __restrict__ mask, int64_t total, int64_t hw, float* __restrict__ ev,
                                      float* __restrict__ loss) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / hw, r = i - n * hw;
    const float* l = logits + n * 4 * hw + r;
    const NigPoint q = nig_point(l[0], l[hw], l[2 * hw], l[3 * hw], label ? label[i] : 0.f);
    float* e = ev + n * 4 * hw + r;
    e[0] = q.mu;
    e[hw] = q.v;
    e[2 * hw] = q.alpha;
    e[3 * hw] = q.beta;
    if (loss && label) {
      const float sq = q.d * q.d;
      float v = q.c * q.T + sq * (2.f * q.alpha + q.v);
      if (mask) v *= mask[i];
      loss[i] = v;
    }
  }
}

// dlogits = d_loss * dloss/dlogits (+ d_ev * dev/dlogits): both upstream gradients optional
__global__ void evidential_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ label,
                                      const float* __restrict__ mask, const float* __restrict__ d_ev,
                                      const float* __restrict__ d_loss, int64_t total, int64_t hw, float* __restrict__ dlogits) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / hw, r = i - n * hw;
    const float* l = logits + n * 4 * hw + r;
    const float l1 = l[hw], l2 = l[2 * hw], l3 = l[3 * hw];
    float g_mu = 0.f, g_v = 0.f, g_a = 0.f, g_b = 0.f;
    if (d_loss && label) {
      const NigPoint q = nig_point(l[0], l1, l2, l3, label[i]);
      const float up = d_loss[i] * (mask ? mask[i] : 1.f);
      const float sq = q.d * q.d, two_a1 = 2.f * q.alpha - 1.f;
      g_mu = up * (-2.f * q.d) * (q.c * two_a1 * q.v + 2.f * q.alpha + q.v);
      g_v = up * (-q.c * q.T / q.v + q.c * (2.f * q.beta + two_a1 * sq) + sq);
      g_a = up * (q.c * (digamma_f(q.alpha - 0.5f) - digamma_f(q.alpha)) * q.T + q.c * 2.f * q.v * sq + 2.f * sq);
      g_b = up * (-q.c * q.T / (2.f * q.beta) + 2.f * q.c * (1.f + q.v));
    }
    if (d_ev) {
      const float* e = d_ev + n * 4 * hw + r;
      g_mu += e[0];
      g_v += e[hw];
      g_a += e[2 * hw];
      g_b += e[3 * hw];
    }
    float* o = dlogits + n * 4 * hw + r;
    o[0] = g_mu;
    o[hw] = g_v * sigmoid_f(l1);
    o[2 * hw] = g_a * sigmoid_f(l2);
    o[3 * hw] = g_b * sigmoid_f(l3);
  }
}

extern "C" int mimo_evidential_forward(const float* logits, const float* label, const float* mask, int32_t n, int64_t hw,
                                       float* ev, float* loss_map, mimo_stream stream) {
  using namespace mimo;
  if (!logits || !ev || n < 0 || hw < 0 || (loss_map && !label)) {
    set_error("mimo_evidential_forward: invalid argument");
    return MIMO_ERR_INVALID;
  }
  const int64_t total = (int64_t)n * hw;
  if (total == 0) return MIMO_OK;
  const int blocks = (int)std::min<int64_t>(ceil_div64(total, 256), 4096);
  hipLaunchKernelGGL(evidential_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, label, mask, total, hw, ev,
                     loss_map);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_evidential_backward(const float* logits, const float* label, const float* mask, const float* d_ev,
                                        const float* d_loss, int32_t n, int64_t hw, float* dlogits, mimo_stream stream) {
  using namespace mimo;
  if (!logits || !dlogits || n < 0 || hw < 0 || (d_loss && !label)) {
    set_error("mimo_evidential_backward: invalid argument");
    return MIMO_ERR_INVALID;
  }
  const int64_t total = (int64_t)n * hw;
  if (total == 0) return MIMO_OK;
  const int blocks = (int)std::min<int64_t>(ceil_div64(total, 256), 4096);
  hipLaunchKernelGGL(evidential_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, logits, label, mask, d_ev, d_loss,
                     total, hw, dlogits);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int32_t step, float grad_scale,
                              mimo_stream stream) {
  using namespace mimo;
  if (!params || !grads || !exp_avg || !exp_avg_sq || n < 0 || step < 1) {
    set_error("mimo_adam_step: invalid argument");
    return MIMO_ERR_INVALID;
  }
  if (n == 0) return MIMO_OK;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const int blocks = (int)std::min<int64_t>(ceil_div64(n / 4 + 1, 256), 2048);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n,
                     lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_adam_step_amp(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                  float beta1, float beta2, float eps, float weight_decay, float* step_dev, float grad_scale,
                                  const float* amp_scale, const float* found_inf, mimo_stream stream) {
  using namespace mimo;
  if (!params || !grads || !exp_avg || !exp_avg_sq || n < 0 || !step_dev) {
    set_error("mimo_adam_step_amp: invalid argument");
    return MIMO_ERR_INVALID;
  }
  hipLaunchKernelGGL(amp_step_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev, found_inf);
  MIMO_KERNEL_CHECK();
  if (n == 0) return MIMO_OK;
  const int blocks = (int)std::min<int64_t>(ceil_div64(n / 4 + 1, 256), 2048);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n,
                     lr, beta1, beta2, eps, weight_decay, 1.f, 1.f, grad_scale, (const float*)step_dev, amp_scale, found_inf);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_uncertainties(const float* p1, const float* p2, int32_t n, int32_t s, int32_t c, int64_t hw,
                                  int32_t loss_kind, float* mean, float* aleatoric, float* epistemic, mimo_stream stream) {
  using namespace mimo;
  if (!p1 || !p2 || !mean || !aleatoric || !epistemic || n < 1 || s < 1 || c < 1 || hw < 1) {
    set_error("mimo_uncertainties: invalid argument");
    return MIMO_ERR_INVALID;
  }
  const int64_t total = (int64_t)n * c * hw;
  const int blocks = (int)std::min<int64_t>(ceil_div64(total, 256), 4096);
  hipLaunchKernelGGL(uncertainty_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p1, p2, n, s, (int64_t)c * hw,
                     loss_kind, mean, aleatoric, epistemic);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

extern "C" int mimo_loss_buffer_step(float* ring, int32_t size, int32_t index, int32_t s, float temperature, const float* loss,
                                     float* weights, float* w_over_s, float* scalars, mimo_stream stream) {
  using namespace mimo;
  if (!ring || !loss || !weights || !w_over_s || !scalars || size < 1 || index < 0 || index >= size || s < 1 || s > 64 ||
      !(temperature > 0.f)) {
    set_error("mimo_loss_buffer_step: invalid argument (1 <= S <= 64, 0 <= index < size, temperature > 0)");
    return MIMO_ERR_INVALID;
  }
  hipLaunchKernelGGL(loss_buffer_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ring, size, index, s, temperature, loss,
                     weights, w_over_s, scalars);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}
