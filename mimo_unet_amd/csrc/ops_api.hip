// Single-operator C-ABI entry points (mimo_op_*): thin drivers around the same kernels the
// plan runs, for the per-kernel parity tests.  They allocate temporaries and synchronise —
// test plumbing, never on the timed path.
#include <algorithm>
#include <vector>

#include "elementwise.h"

using namespace mimo;

namespace {

struct Temp {
  std::vector<void*> ptrs;
  ~Temp() {
    for (void* p : ptrs) (void)hipFree(p);
  }
  template <typename T>
  T* get(size_t count) {
    void* q = nullptr;
    if (hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return nullptr;
    (void)hipMemset(q, 0, std::max<size_t>(count, 1) * sizeof(T));
    ptrs.push_back(q);
    return static_cast<T*>(q);
  }
  int* ints(const std::vector<int>& v) {
    int* p = get<int>(v.size());
    if (p) (void)hipMemcpy(p, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice);
    return p;
  }
};

std::vector<int> ident_map(int padded, int logical) {
  std::vector<int> m(padded);
  for (int i = 0; i < padded; ++i) m[i] = i < logical ? i : -1;
  return m;
}

__global__ void colsum_naive_kernel(const float* __restrict__ x, int64_t rows, int ld, int C, float* __restrict__ out) {
  const int c = blockIdx.x;
  __shared__ double red[256];
  double s = 0.0;
  for (int64_t r = threadIdx.x; r < rows; r += blockDim.x) s += (double)x[r * ld + c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0 && c < C) out[c] = (float)red[0];
}

__global__ void sums_to_stats_kernel(const double* __restrict__ sums, int chunks, int cout_pad, int cout,
                                     double* __restrict__ stats) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cout) return;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < chunks; ++k) {
    s1 += sums[(size_t)k * 2 * cout_pad + c];
    s2 += sums[(size_t)k * 2 * cout_pad + cout_pad + c];
  }
  stats[c] = s1;
  stats[cout + c] = s2;
}

// fp32 <-> 16-bit storage conversions, so that the fp32 test interface can drive the 16-bit storage kernels
template <typename T>
__global__ void to16_kernel(const float* __restrict__ s, T* __restrict__ d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = (T)s[i];
}
template <typename T>
__global__ void from16_kernel(const T* __restrict__ s, float* __restrict__ d, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = (float)s[i];
}
int to16(const float* s, void* d, int64_t n, bool f16, hipStream_t st) {
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
  if (f16)
    hipLaunchKernelGGL(to16_kernel<_Float16>, dim3(blocks), dim3(256), 0, st, s, (_Float16*)d, n);
  else
    hipLaunchKernelGGL(to16_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, s, (__bf16*)d, n);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}
int from16(const void* s, float* d, int64_t n, bool f16, hipStream_t st) {
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
  if (f16)
    hipLaunchKernelGGL(from16_kernel<_Float16>, dim3(blocks), dim3(256), 0, st, (const _Float16*)s, d, n);
  else
    hipLaunchKernelGGL(from16_kernel<__bf16>, dim3(blocks), dim3(256), 0, st, (const __bf16*)s, d, n);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}
bool is_mixed(int precision) { return precision == MIMO_PREC_BF16_MIXED || precision == MIMO_PREC_FP16_MIXED; }

}  // namespace

extern "C" {

int mimo_op_conv3x3_forward(const float* x, const float* w, const float* bias, float* z, double* stats, int32_t n,
                            int32_t h, int32_t wd, int32_t cin, int32_t cin_p, int32_t cout, int32_t cout_p,
                            int32_t precision, mimo_stream stream) {
  hipStream_t st = (hipStream_t)stream;
  Temp t;
  const int cout_pad = conv3x3_cout_pad(cout);
  // 16-bit storage modes: the fp32 tensors of this interface are rounded into 16-bit buffers, the storage-mode kernels
  // run on those, the result is widened again (exact)
  const bool mixed = is_mixed(precision), f16s = precision == MIMO_PREC_FP16_MIXED;
  const bool split = precision == MIMO_PREC_SPLIT16 || precision == MIMO_PREC_BF16 || mixed;
  const bool bf16 = precision == MIMO_PREC_BF16 || precision == MIMO_PREC_BF16_MIXED;
  float* wf = t.get<float>((size_t)9 * cout_pad * cin_p);
  void* wpk = t.get<uint16_t>((size_t)ceil_div(cin_p, 32) * 9 * cout_pad * 64);
  float* bp = t.get<float>(cout_pad);
  int* rm = t.ints(ident_map(cout_pad, cout));
  int* cm = t.ints(ident_map(cin_p, cin));
  const int rows_cap = std::max(std::max(conv3x3_stat_rows(n, h, wd), conv3x3_ws_stat_rows(n, h, wd)), 2048);
  float* partial = t.get<float>((size_t)rows_cap * 2 * cout_pad);
  double* sums = t.get<double>((size_t)kMaxChunks * 2 * cout_pad);
  if (!wf || !bp || !rm || !cm || !partial || !sums) {
    set_error("mimo_op_conv3x3_forward: allocation failed");
    return MIMO_ERR_HIP;
  }
  MIMO_TRY(pack_weights_launch(w, wf, cout, cin, cout_pad, cin_p, rm, cm, 0, st));
  const int fmode = mixed ? (f16s ? 6 : 4) : (bf16 ? 2 : 1);
  // split16: the decomposition the plan would pick for this geometry (conv_wide.hip or conv_bf16x3.hip)
  const int wide = (precision == MIMO_PREC_SPLIT16 || mixed) ? conv3x3_wide_rows(fmode, n, cin_p, cout_p, h, wd) : 0;
  const int pair = (split && !wide) ? conv3x3_pair_tail(fmode, cin_p, h, wd) : 0;
  if (wide) {
    wpk = t.get<uint16_t>(conv3x3_wide_weight_elems(cin_p, wide));
    if (!wpk) {
      set_error("mimo_op_conv3x3_forward: allocation failed");
      return MIMO_ERR_HIP;
    }
    MIMO_TRY(pack_weights_wide_launch(w, wpk, bf16 ? 0 : 1, cout, cin, wide, cin_p, rm, cm, cout_pad, 0, st, mixed ? 1 : 0));
  } else if (split)
    MIMO_TRY(pack_weights_bf16x3_launch(w, wpk, bf16 ? 0 : 1, cout, cin, cout_pad, cin_p, rm, cm, 0, st, pair));
  if (bias) MIMO_HIP_CHECK(hipMemcpyAsync(bp, bias, cout * sizeof(float), hipMemcpyDeviceToDevice, st));
  const int64_t nx = (int64_t)n * h * wd * cin_p, nz = (int64_t)n * h * wd * cout_p;
  uint16_t *x16 = nullptr, *z16 = nullptr;
  if (mixed) {
    x16 = t.get<uint16_t>(nx);
    z16 = t.get<uint16_t>(nz);
    if (!x16 || !z16) {
      set_error("mimo_op_conv3x3_forward: allocation failed");
      return MIMO_ERR_HIP;
    }
    MIMO_TRY(to16(x, x16, nx, f16s, st));
  }
  ConvLaunch a;
  a.x = mixed ? reinterpret_cast<const float*>(x16) : x;
  a.y = mixed ? reinterpret_cast<float*>(z16) : z;
  a.w = wf;
  a.bias = bp;
  a.stats = stats ? partial : nullptr;
  a.N = n;
  a.Hi = a.Ho = h;
  a.Wi = a.Wo = wd;
  a.ldx = cin_p;
  a.cin_p = cin_p;
  a.ldy = cout_p;
  a.cout_pad = cout_pad;
  a.cout_store = cout_p;
  a.off = 1;
  a.wpk = wpk;
  a.pair = pair;
  a.wide = wide;
  int rows = 0;
  // (an image convolution — <= 4 input channels in an 8-channel pixel — runs on the plain-FMA kernel, as in the plan)
  if (!mixed && cin_p < 16 && conv3x3_thin_ok(cin, cout_p))
    MIMO_TRY(conv3x3_thin_launch(a, cin, &rows, st));
  else if (split) {
    // (with the K split the plan would use for this geometry, conv3x3_ksplit: scratch for its partial slabs)
    const size_t kfl = (mixed || wide) ? 0 : conv3x3_ksplit_scratch(fmode, n, cin_p, cout_pad, h, wd, cout_p);
    float* kpart = kfl ? t.get<float>(kfl) : nullptr;
    MIMO_TRY(conv3x3_bf16x3_launch_k(a, fmode, &rows, st, kpart, kfl));
  } else
    MIMO_TRY(conv3x3_launch(a, &rows, st));
  if (mixed) MIMO_TRY(from16(z16, z, nz, f16s, st));
  if (stats) {
    int chunks = 0;
    MIMO_TRY(rowsum_launch(partial, rows, 2 * cout_pad, sums, &chunks, st));
    hipLaunchKernelGGL(sums_to_stats_kernel, dim3(ceil_div(cout, 64)), dim3(64), 0, st, sums, chunks, cout_pad, cout, stats);
    MIMO_KERNEL_CHECK();
  }
  MIMO_HIP_CHECK(hipStreamSynchronize(st));
  return MIMO_OK;
}

int mimo_op_conv3x3_dgrad(const float* dz, const float* w, float* dx, int32_t n, int32_t h, int32_t wd, int32_t cin,
                          int32_t cin_p, int32_t cout, int32_t cout_p, int32_t precision, mimo_stream stream) {
  hipStream_t st = (hipStream_t)stream;
  Temp t;
  const int rows_pad = conv3x3_cout_pad(cin_p);
  const bool mixed = is_mixed(precision), f16s = precision == MIMO_PREC_FP16_MIXED;
  const bool split = precision == MIMO_PREC_SPLIT16 || precision == MIMO_PREC_BF16 || mixed;
  const bool bf16 = precision == MIMO_PREC_BF16;
  float* wdp = t.get<float>((size_t)9 * rows_pad * cout_p);
  void* wpk = t.get<uint16_t>((size_t)ceil_div(cout_p, 32) * 9 * rows_pad * 64);
  int* rm = t.ints(ident_map(rows_pad, cin));
  int* cm = t.ints(ident_map(cout_p, cout));
  float* dxpad = t.get<float>((size_t)n * (h + 2) * (wd + 2) * cin_p);
  if (!wdp || !rm || !cm || !dxpad) {
    set_error("mimo_op_conv3x3_dgrad: allocation failed");
    return MIMO_ERR_HIP;
  }
  MIMO_TRY(pack_weights_launch(w, wdp, cout, cin, rows_pad, cout_p, rm, cm, 1, st));
  const int dmode = mixed ? (f16s ? 7 : 5) : (bf16 ? 3 : 0);
  const int wide = (precision == MIMO_PREC_SPLIT16 || mixed) ? conv3x3_wide_rows(dmode, n, cout_p, cin_p, h + 2, wd + 2) : 0;
  const int pair = (split && !wide) ? conv3x3_pair_tail(dmode, cout_p, h + 2, wd + 2) : 0;
  if (wide) {
    wpk = t.get<uint16_t>(conv3x3_wide_weight_elems(cout_p, wide));
    if (!wpk) {
      set_error("mimo_op_conv3x3_dgrad: allocation failed");
      return MIMO_ERR_HIP;
    }
    MIMO_TRY(pack_weights_wide_launch(w, wpk, f16s ? 1 : 0, cout, cin, wide, cout_p, rm, cm, rows_pad, 1, st, mixed ? 1 : 0));
  } else if (split)
    MIMO_TRY(pack_weights_bf16x3_launch(w, wpk, f16s ? 1 : 0, cout, cin, rows_pad, cout_p, rm, cm, 1, st, pair));
  const float* dz_in = dz;
  const int64_t ndz = (int64_t)n * h * wd * cout_p, ndx = (int64_t)n * h * wd * cin_p;
  uint16_t* dx16 = nullptr;
  if (mixed) {  // plain 16-bit dz in, 16-bit padded-domain gradient out, folded in 16-bit storage, widened at the end
    uint16_t* dz16 = t.get<uint16_t>(ndz);
    dx16 = t.get<uint16_t>(ndx);
    if (!dz16 || !dx16) {
      set_error("mimo_op_conv3x3_dgrad: allocation failed");
      return MIMO_ERR_HIP;
    }
    MIMO_TRY(to16(dz, dz16, ndz, f16s, st));
    dz_in = reinterpret_cast<const float*>(dz16);
  } else if (split) {  // the bf16-pair kernel reads dz in split storage (see elementwise.h split_pairs_launch)
    float* dzs = t.get<float>((size_t)n * h * wd * cout_p);
    if (!dzs) {
      set_error("mimo_op_conv3x3_dgrad: allocation failed");
      return MIMO_ERR_HIP;
    }
    MIMO_TRY(split_pairs_launch(dz, dzs, (int64_t)n * h * wd, cout_p, st));
    dz_in = dzs;
  }
  ConvLaunch a;
  a.x = dz_in;
  a.y = dxpad;
  a.w = wdp;
  a.bias = nullptr;
  a.stats = nullptr;
  a.N = n;
  a.Hi = h;
  a.Wi = wd;
  a.ldx = cout_p;
  a.cin_p = cout_p;
  a.Ho = h + 2;
  a.Wo = wd + 2;
  a.ldy = cin_p;
  a.cout_pad = rows_pad;
  a.cout_store = cin_p;
  a.off = 2;
  a.wpk = wpk;
  a.pair = pair;
  a.wide = wide;
  if (split) {
    const size_t kfl = (mixed || wide) ? 0 : conv3x3_ksplit_scratch(dmode, n, cout_p, rows_pad, h + 2, wd + 2, cin_p);
    float* kpart = kfl ? t.get<float>(kfl) : nullptr;
    MIMO_TRY(conv3x3_bf16x3_launch_k(a, dmode, nullptr, st, kpart, kfl));
  } else
    MIMO_TRY(conv3x3_launch(a, nullptr, st));
  if (mixed) {
    MIMO_TRY(fold_slice_launch(dxpad, f16s ? ST_F16 : ST_BF16, cin_p, 0, dx16, cin_p, n, h, wd, cin_p, 0, st));
    MIMO_TRY(from16(dx16, dx, ndx, f16s, st));
  } else {
    MIMO_TRY(fold_slice_launch(dxpad, ST_F32, cin_p, 0, dx, cin_p, n, h, wd, cin_p, 0, st));
  }
  MIMO_HIP_CHECK(hipStreamSynchronize(st));
  return MIMO_OK;
}

int mimo_op_conv3x3_wgrad(const float* x, const float* dz, float* dw, float* dbias, int32_t n, int32_t h, int32_t wd,
                          int32_t cin, int32_t cin_p, int32_t cout, int32_t cout_p, int32_t precision,
                          mimo_stream stream) {
  hipStream_t st = (hipStream_t)stream;
  Temp t;
  const bool mixed = is_mixed(precision), f16s = precision == MIMO_PREC_FP16_MIXED;
  const bool split = precision == MIMO_PREC_SPLIT16 || precision == MIMO_PREC_BF16 || mixed;
  const bool bf16 = precision == MIMO_PREC_BF16;
  if (!mixed && cin_p < 16 && wgrad_thin_ok(cin, cout_p, n, h, wd)) {  // the image convolution's kernel, as in the plan
    float* part = t.get<float>(wgrad_thin_scratch(cin, cout_p));
    if (!part) {
      set_error("mimo_op_conv3x3_wgrad: allocation failed");
      return MIMO_ERR_HIP;
    }
    MIMO_TRY(wgrad_thin_launch(x, cin_p, dz, cout_p, n, h, wd, cin, cout, cout_p, part, dw, st));
    if (dbias) {
      hipLaunchKernelGGL(colsum_naive_kernel, dim3(cout), dim3(256), 0, st, dz, (int64_t)n * h * wd, cout_p, cout, dbias);
      MIMO_KERNEL_CHECK();
    }
    MIMO_HIP_CHECK(hipStreamSynchronize(st));
    return MIMO_OK;
  }
  WgradLaunch a;
  a.x = x;
  a.dz = dz;
  a.N = n;
  a.H = h;
  a.W = wd;
  a.ldx = cin_p;
  a.lddz = cout_p;
  a.cin_p = cin_p;
  a.cout_p = cout_p;
  if (split) {
    int CI, CO;
    wgrad_split_tiles(cin_p, cout_p, &CI, &CO);
    a.cin_pad = round_up(cin_p, CI);
    a.cout_pad = round_up(cout_p, CO);
    a.splits = wgrad_split_pick_splits(n, h, wd, a.cin_pad, a.cout_pad, CI, CO, mixed ? (f16s ? 2 : 1) : 0);
  } else {
    a.cin_pad = round_up(cin_p, 32);
    a.cout_pad = round_up(cout_p, 32);
    a.splits = wgrad_pick_splits(n, h, wd, a.cin_pad, a.cout_pad);
  }
  a.partial = t.get<float>((size_t)(a.splits + a.splits / 8 + 2) * 9 * a.cin_pad * a.cout_pad);
  int* cm = t.ints(ident_map(cin_p, cin));
  if (!a.partial || !cm) {
    set_error("mimo_op_conv3x3_wgrad: allocation failed");
    return MIMO_ERR_HIP;
  }
  if (mixed) {  // activations and dz as plain 16-bit NHWC tensors
    const int64_t nx = (int64_t)n * h * wd * cin_p, ndz = (int64_t)n * h * wd * cout_p;
    uint16_t *x16 = t.get<uint16_t>(nx), *dz16 = t.get<uint16_t>(ndz);
    if (!x16 || !dz16) {
      set_error("mimo_op_conv3x3_wgrad: allocation failed");
      return MIMO_ERR_HIP;
    }
    MIMO_TRY(to16(x, x16, nx, f16s, st));
    MIMO_TRY(to16(dz, dz16, ndz, f16s, st));
    a.x = reinterpret_cast<const float*>(x16);
    a.dz = reinterpret_cast<const float*>(dz16);
    a.np = 1;
    a.store = f16s ? 2 : 1;
    MIMO_TRY(wgrad_split_launch(a, st));
  } else if (split) {  // split storage of dz, as the BatchNorm-backward kernel writes it in the plan
    float* dzs = t.get<float>((size_t)n * h * wd * cout_p);
    if (!dzs) {
      set_error("mimo_op_conv3x3_wgrad: allocation failed");
      return MIMO_ERR_HIP;
    }
    // as in the plan: two fp16 MFMAs per product where the geometry has that kernel, with max |dz| next to the split
    float* amax = (!bf16 && wgrad_split_has_np2(cin_p, cout_p)) ? t.get<float>(kDzMaxSlots) : nullptr;
    int amax_n = 0;
    MIMO_TRY(split_pairs_launch(dz, dzs, (int64_t)n * h * wd, cout_p, st, amax, &amax_n));
    a.dz = dzs;
    a.np = bf16 ? 1 : amax ? 2 : 3;
    a.dz_absmax = amax;
    a.dz_absmax_n = amax_n;
    MIMO_TRY(wgrad_split_launch(a, st));
  } else {
    MIMO_TRY(wgrad_launch(a, st));
  }
  MIMO_TRY(wgrad_reduce_launch(a.partial, a.splits, a.cin_pad, a.cout_pad, cm, cin_p, cin, cout, dw, st, a.dz_absmax, a.dz_absmax_n));
  if (dbias) {
    hipLaunchKernelGGL(colsum_naive_kernel, dim3(cout), dim3(256), 0, st, dz, (int64_t)n * h * wd, cout_p, cout, dbias);
    MIMO_KERNEL_CHECK();
  }
  MIMO_HIP_CHECK(hipStreamSynchronize(st));
  return MIMO_OK;
}

int mimo_op_maxpool2x2(const float* x, float* y, int32_t n, int32_t h, int32_t w, int32_t c_p, mimo_stream stream) {
  MIMO_TRY(maxpool_fwd_launch(x, ST_F32, c_p, n, h, w, c_p, y, c_p, (hipStream_t)stream));
  return MIMO_OK;
}

int mimo_op_upsample_cat(const float* skip, const float* low, float* out, int32_t n, int32_t hs, int32_t ws, int32_t cs_p,
                         int32_t hl, int32_t wl, int32_t cl_p, mimo_stream stream) {
  MIMO_TRY(upcat_fwd_launch(skip, ST_F32, cs_p, cs_p, low, cl_p, cl_p, n, hs, ws, hl, wl, out, (hipStream_t)stream));
  return MIMO_OK;
}

}  // extern "C"
