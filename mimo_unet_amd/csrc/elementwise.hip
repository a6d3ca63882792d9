// Bandwidth-class kernels of the MIMO U-Net path for gfx950: everything that is not a 3x3
// convolution.  One float4 (4 channels of one pixel) per thread per step, fully coalesced
// NHWC accesses, per-thread register accumulation + LDS block reduction + per-workgroup
// partial rows (deterministic two-level reduction, no float atomics).
//
// Reference operators replaced (relative to /root/reference):
//   BatchNorm2d / ReLU / Dropout2d ........ mimo/models/mimo_components/components.py:24-29
//   MaxPool2d(2) .......................... components.py:48
//   Upsample(bilinear, align_corners) + F.pad + cat ... components.py:78,110-119
//   OutConv 1x1 ........................... components.py:123-129
//   LaplaceNLL / GaussianNLL .............. mimo/losses.py:47-79,132-164
//   apply_input_transform gather .......... mimo/models/utils.py:38-48
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>

#include "elementwise.h"

namespace mimo {

// ---------------------------------------------------------------------------------------
// thread mapping: 256 threads = QB channel-quads x PPI pixels; quad fixed per thread
// ---------------------------------------------------------------------------------------
struct PQ {
  int q;       // channel quad handled by this thread
  int ql, pl;  // position inside the workgroup
  int QB, PPI;
  int p, pstep;  // first pixel, pixel stride
  bool active;
};

// xcd_bands: workgroups are dealt round-robin to the 8 XCDs (each with a private L2); give every XCD one
// contiguous band of the pixels in flight, so that gathers which re-read neighbouring rows (bilinear
// backward: every gradient row feeds two input rows) find them in their own L2.
// T = threads per workgroup (256 everywhere but the BatchNorm-backward reduction, which runs 1024)
template <int T = 256>
__device__ __forceinline__ PQ pixquad(int Cv, bool xcd_bands = false) {
  PQ r;
  r.QB = Cv < T ? Cv : T;
  r.PPI = T / r.QB;
  r.ql = threadIdx.x % r.QB;
  r.pl = threadIdx.x / r.QB;
  r.q = blockIdx.y * r.QB + r.ql;
  r.active = r.pl < r.PPI && r.q < Cv;
  int bx = blockIdx.x;
  if (xcd_bands && (gridDim.x & 7) == 0) bx = (bx & 7) * (gridDim.x >> 3) + (bx >> 3);
  r.p = bx * r.PPI + r.pl;
  r.pstep = gridDim.x * r.PPI;
  return r;
}

// Grid caps of the grid-stride kernels, scanned per kernel on the cfg3 step (rocprofv3 kernel trace; 256 CUs):
// the two BatchNorm-backward passes run best with exactly the 7 workgroups per CU their registers allow resident
// (1024 -> 1792 blocks: reduce 1.92 -> 1.63 ms, apply 2.38 -> 2.21 ms per step; 2048 already spills into a second,
// partial round), BatchNorm + ReLU forward with 6 per CU, the 16-tap bilinear backward with many short
// workgroups (4096 -> 16384: 0.67 -> 0.55 ms), the rest at 4096-8192.
constexpr int kBlocksBnRelu = 1536, kBlocksBnBwd = 1792, kBlocksUpBwd = 16384, kBlocksPoolBwd = 8192;
// The BatchNorm-backward REDUCTION runs the same 7168 waves as 448 workgroups of 1024 threads: a quarter of the partial
// rows for the column-sum launch that follows (12.6 -> ~5 us, 24 times per step, at every batch size)
constexpr int kBnReduceThreads = 1024, kBlocksBnReduce = kBlocksBnBwd / 4;
// (Round 6 measured fewer, longer threads on the small tensors of a 4-image step — at least 2 / 4 / 8 pixels per thread, so
// that the 4-6 float4 of channel constants a thread loads are not most of its memory instructions: the BatchNorm-backward
// passes got SLOWER, 0.78 -> 0.80 / 0.91 / 1.10 ms per step; they want the threads.  profiles/r06/exp/ew_min_iters.txt)
static dim3 pq_grid(int Cv, int64_t P, int max_blocks = kEwMaxBlocks, int T = 256) {
  const int QB = Cv < T ? Cv : T;
  const int PPI = T / QB;
  int64_t gx = ceil_div64(P, PPI);
  if (gx > max_blocks) gx = max_blocks;
  if (gx < 1) gx = 1;
  return dim3((unsigned)gx, (unsigned)ceil_div(Cv, QB));
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// 16-bit storage (mimo_precision *_MIXED): four channels = one 8-byte access, arithmetic stays fp32
typedef __bf16 bf16x4_st __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4_st __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const __bf16* p) {
  const bf16x4_st v = *reinterpret_cast<const bf16x4_st*>(p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void st4(__bf16* p, float4 v) {
  bf16x4_st r;
  r[0] = (__bf16)v.x;
  r[1] = (__bf16)v.y;
  r[2] = (__bf16)v.z;
  r[3] = (__bf16)v.w;
  *reinterpret_cast<bf16x4_st*>(p) = r;
}
__device__ __forceinline__ float4 ld4(const _Float16* p) {
  const f16x4_st v = *reinterpret_cast<const f16x4_st*>(p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void st4(_Float16* p, float4 v) {
  f16x4_st r;
  r[0] = (_Float16)v.x;
  r[1] = (_Float16)v.y;
  r[2] = (_Float16)v.z;
  r[3] = (_Float16)v.w;
  *reinterpret_cast<f16x4_st*>(p) = r;
}
// Launch wrappers take untyped pointers plus a StoreType (elementwise.h); BODY sees the element type as T.
#define MIMO_ST_DISPATCH(DT, T, ...) \
  switch (DT) {                      \
    case ST_BF16: {                  \
      typedef __bf16 T;              \
      __VA_ARGS__;                   \
    } break;                         \
    case ST_F16: {                   \
      typedef _Float16 T;            \
      __VA_ARGS__;                   \
    } break;                         \
    default: {                       \
      typedef float T;               \
      __VA_ARGS__;                   \
    } break;                         \
  }
// (z type, activation type) pairs of the BatchNorm kernels: z is fp32 where the convolution that wrote it runs on the
// fp32 kernel family (the 2..4-channel image convolution) even in the 16-bit storage modes
#define MIMO_ST_DISPATCH2(DTZ, DTA, TZ, TA, ...) \
  if ((DTZ) == ST_F32) {                         \
    typedef float TZ;                            \
    MIMO_ST_DISPATCH(DTA, TA, __VA_ARGS__)       \
  } else if ((DTA) == ST_BF16) {                 \
    typedef __bf16 TZ;                           \
    typedef __bf16 TA;                           \
    __VA_ARGS__;                                 \
  } else {                                       \
    typedef _Float16 TZ;                         \
    typedef _Float16 TA;                         \
    __VA_ARGS__;                                 \
  }

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4max(float4 a, float4 b) {
  return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}

// sum the accumulators of the threads that share a channel quad; result valid for pl == 0 (red: one float4 per thread).
// Fixed summation order (deterministic); two levels when many threads share a quad (the 1024-thread reduction).
__device__ __forceinline__ float4 quad_block_sum(float4 v, const PQ& t, float4* red) {
  __syncthreads();
  red[threadIdx.x] = t.active ? v : f4zero();
  __syncthreads();
  float4 s = f4zero();
  if (t.PPI <= 32) {
    if (t.pl == 0)
      for (int j = 0; j < t.PPI; ++j) s = f4add(s, red[j * t.QB + t.ql]);
    return s;
  }
  const bool lead = t.pl < 16 && t.pl < t.PPI;
  if (lead)
    for (int j = t.pl; j < t.PPI; j += 16) s = f4add(s, red[j * t.QB + t.ql]);
  __syncthreads();
  if (lead) red[threadIdx.x] = s;
  __syncthreads();
  s = f4zero();
  if (t.pl == 0)
    for (int j = 0; j < 16 && j < t.PPI; ++j) s = f4add(s, red[j * t.QB + t.ql]);
  return s;
}

// gradient on the reflect-padded domain folded back onto the image (transpose of reflect pad):
// pad row -1 lands on row 1, pad row H on row H-2 (same for columns).
template <typename T>
__device__ __forceinline__ float4 fold_read(const T* dxpad, int ldp, int n, int y, int x, int H, int W, int ch) {
  const T* base = dxpad + (size_t)n * (H + 2) * (W + 2) * ldp + ch;
  float4 s = ld4(base + ((size_t)(y + 1) * (W + 2) + (x + 1)) * ldp);  // interior pixels: this one load
  const bool ya = y == 1, yb = y == H - 2, xa = x == 1, xb = x == W - 2;
  if (ya | yb | xa | xb) {  // rows 1 / H-2 and columns 1 / W-2 also receive the reflected border
    const int ry[3] = {y, -1, H}, rx[3] = {x, -1, W};
    const bool vy[3] = {true, ya, yb}, vx[3] = {true, xa, xb};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (i + j > 0 && vy[i] && vx[j]) s = f4add(s, ld4(base + ((size_t)(ry[i] + 1) * (W + 2) + (rx[j] + 1)) * ldp));
  }
  return s;
}

// (n, y, x) of a pixel index that advances by a fixed stride: two integer divisions once per thread
// instead of two per visited pixel (they cost more VALU than the 48-64 bytes the pixel moves)
struct PixIter {
  int n, y, x, dn, dy, dx;
};
__device__ __forceinline__ PixIter pix_iter(int p0, int pstep, int H, int W) {
  PixIter it;
  const int HW = H * W;
  it.n = p0 / HW;
  int r = p0 - it.n * HW;
  it.y = r / W;
  it.x = r - it.y * W;
  it.dn = pstep / HW;
  r = pstep - it.dn * HW;
  it.dy = r / W;
  it.dx = r - it.dy * W;
  return it;
}
__device__ __forceinline__ void pix_next(PixIter& it, int H, int W) {
  it.x += it.dx;
  it.y += it.dy;
  it.n += it.dn;
  if (it.x >= W) {
    it.x -= W;
    ++it.y;
  }
  if (it.y >= H) {
    it.y -= H;
    ++it.n;
  }
}

__device__ __forceinline__ void pix_prev(PixIter& it, int H, int W) {
  it.x -= it.dx;
  it.y -= it.dy;
  it.n -= it.dn;
  if (it.x < 0) {
    it.x += W;
    --it.y;
  }
  if (it.y < 0) {
    it.y += H;
    --it.n;
  }
}

// ---------------------------------------------------------------------------------------
// two-level column reduction
// ---------------------------------------------------------------------------------------
__global__ void rowsum_kernel(const float* __restrict__ partial, int rows, int cols, int rows_per_chunk,
                              double* __restrict__ sums) {
  __shared__ double red[256];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(rows, r0 + rows_per_chunk);
  double s = 0.0;
  if (col < cols)
    for (int r = r0 + rl; r < r1; r += 4) s += (double)partial[(size_t)r * cols + col];
  red[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && col < cols) sums[(size_t)blockIdx.y * cols + col] = red[cl] + red[64 + cl] + red[128 + cl] + red[192 + cl];
}

int rowsum_launch(const float* partial, int rows, int cols, double* sums, int* chunks, hipStream_t s) {
  int nch = ceil_div(rows, 32);
  if (nch > kMaxChunks) nch = kMaxChunks;
  if (nch < 1) nch = 1;
  const int rpc = ceil_div(rows, nch);
  nch = ceil_div(rows, rpc);
  *chunks = nch;
  hipLaunchKernelGGL(rowsum_kernel, dim3(ceil_div(cols, 64), nch), dim3(256), 0, s, partial, rows, cols, rpc, sums);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// The finalize kernels run 256 threads per 64 columns: four groups each add every fourth chunk row (the
// serial walk over up to kMaxChunks rows was most of their ~8 us), the group totals are added in fixed order.
// Every thread of the workgroup must call this; the total is returned to all of them.
__device__ __forceinline__ double chunk_total(const double* __restrict__ col, size_t stride, int chunks, bool valid,
                                              double* red) {
  double s = 0.0;
  if (valid)
    for (int k = threadIdx.x >> 6; k < chunks; k += 4) s += col[(size_t)k * stride];
  __syncthreads();
  red[threadIdx.x] = s;
  __syncthreads();
  const int cl = threadIdx.x & 63;
  return (red[cl] + red[64 + cl]) + (red[128 + cl] + red[192 + cl]);
}

// two columns in one walk (the BatchNorm finalize kernels need sum and sum of squares): one barrier pair
__device__ __forceinline__ void chunk_total2(const double* __restrict__ col_a, const double* __restrict__ col_b, size_t stride,
                                             int chunks, bool valid, double* red /*[512]*/, double* ta, double* tb) {
  double sa = 0.0, sb = 0.0;
  if (valid)
    for (int k = threadIdx.x >> 6; k < chunks; k += 4) {
      sa += col_a[(size_t)k * stride];
      sb += col_b[(size_t)k * stride];
    }
  __syncthreads();
  red[threadIdx.x] = sa;
  red[256 + threadIdx.x] = sb;
  __syncthreads();
  const int cl = threadIdx.x & 63;
  *ta = (red[cl] + red[64 + cl]) + (red[128 + cl] + red[192 + cl]);
  *tb = (red[256 + cl] + red[320 + cl]) + (red[384 + cl] + red[448 + cl]);
}

// ---- one-launch column sums (deterministic): one workgroup of 1024 threads = 64 columns x 16 row groups walks the
// fp32 partial rows of its 64 columns in double, eight independent loads in flight per thread, then runs the
// finalize arithmetic — one launch instead of the rowsum + finalize pair.  (Spreading the rows over several
// workgroups with a last-workgroup ticket was measured SLOWER, 24 vs 13 us: the device-scope fence it needs writes
// back / invalidates the whole L2 on this multi-XCD part.)  `scratch` / `tickets` are unused, kept for the signature.
constexpr int kColsumThreads = 1024;
static int colsum_chunks(int) { return 1; }

__device__ __forceinline__ bool grid_colsum2(const float* __restrict__ partial, int rows, size_t stride, int col_a, int col_b,
                                             bool valid, bool two, double* red /*[2048]*/, double* __restrict__, int,
                                             int* __restrict__, double* ta, double* tb) {
  const int rg = threadIdx.x >> 6;
  double sa = 0.0, sb = 0.0;
  if (valid) {
    const float* pa = partial + col_a;
    const float* pb = partial + col_b;
    int r = rg;
    if (two) {
      for (; r + 48 < rows; r += 64) {
        float a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          a[k] = pa[(size_t)(r + 16 * k) * stride];
          b[k] = pb[(size_t)(r + 16 * k) * stride];
        }
        sa += ((double)a[0] + (double)a[1]) + ((double)a[2] + (double)a[3]);
        sb += ((double)b[0] + (double)b[1]) + ((double)b[2] + (double)b[3]);
      }
    } else {
      for (; r + 112 < rows; r += 128) {
        float a[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = pa[(size_t)(r + 16 * k) * stride];
        sa += (((double)a[0] + (double)a[1]) + ((double)a[2] + (double)a[3])) +
              (((double)a[4] + (double)a[5]) + ((double)a[6] + (double)a[7]));
      }
    }
    for (; r < rows; r += 16) {
      sa += (double)pa[(size_t)r * stride];
      if (two) sb += (double)pb[(size_t)r * stride];
    }
  }
  __syncthreads();
  red[threadIdx.x] = sa;
  red[1024 + threadIdx.x] = sb;
  __syncthreads();
  const int cl = threadIdx.x & 63;
  double a = 0.0, b = 0.0;
  if (threadIdx.x < 64) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      a += red[k * 64 + cl];
      b += red[1024 + k * 64 + cl];
    }
  }
  *ta = a;
  *tb = b;
  return true;
}

// out[c] = sum over rows of partial[r][c]  (conv bias gradient)
__global__ __launch_bounds__(kColsumThreads) void colsum_vec_kernel(const float* __restrict__ partial, int rows, int cols,
                                                                    int C, float* __restrict__ out,
                                                                    double* __restrict__ scratch, int scratch_cols,
                                                                    int* __restrict__ tickets) {
  __shared__ double red[2048];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  double s, unused;
  if (!grid_colsum2(partial, rows, (size_t)cols, c, c, c < C, false, red, scratch, scratch_cols, tickets, &s, &unused)) return;
  if (c < C && threadIdx.x < 64) out[c] = (float)s;
}

int colsum_vec_launch(const float* partial, int rows, int cols, int C, float* out, const ColsumScratch& cs, hipStream_t st) {
  const int groups = ceil_div(C, 64);
  hipLaunchKernelGGL(colsum_vec_kernel, dim3(groups, colsum_chunks(rows)), dim3(kColsumThreads), 0, st, partial, rows, cols, C,
                     out, cs.sums, groups * 64, cs.tickets);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

__global__ void vec_finalize_kernel(const double* __restrict__ sums, int chunks, int cols, int C, float* __restrict__ out) {
  __shared__ double red[256];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const double s = chunk_total(sums + c, cols, chunks, c < C, red);
  if (c < C && threadIdx.x < 64) out[c] = (float)s;
}

int vec_finalize_launch(const double* sums, int chunks, int cols, int C, float* out, hipStream_t st) {
  hipLaunchKernelGGL(vec_finalize_kernel, dim3(ceil_div(C, 64)), dim3(256), 0, st, sums, chunks, cols, C, out);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// ---------------------------------------------------------------------------------------
// layout conversion
// ---------------------------------------------------------------------------------------
__global__ void pack_input_kernel(const float* __restrict__ x, int64_t stride_n, int64_t stride_s,
                                  const int64_t* __restrict__ perm, int s, int N, int C, int HW,
                                  float* __restrict__ out, int cp) {
  const int64_t total = (int64_t)N * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / HW);
    const int yx = (int)(i - (int64_t)n * HW);
    const int64_t src_n = perm ? perm[(int64_t)s * N + n] : n;
    const float* src = x + src_n * stride_n + (int64_t)s * stride_s + yx;
    float* dst = out + i * cp;
    for (int c = 0; c < cp; ++c) dst[c] = c < C ? src[(int64_t)c * HW] : 0.f;
  }
}

int pack_input_launch(const float* x, int64_t stride_n, int64_t stride_s, const int64_t* perm, int s, int N, int C,
                      int H, int W, float* out, int cp, hipStream_t st) {
  const int64_t total = (int64_t)N * H * W;
  const int blocks = (int)std::min<int64_t>(ceil_div64(total, 256), 4096);
  hipLaunchKernelGGL(pack_input_kernel, dim3(blocks), dim3(256), 0, st, x, stride_n, stride_s, perm, s, N, C, H * W, out, cp);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

template <typename T>
__global__ void unpack_dx_kernel(const T* __restrict__ dxpad, int ldp, int N, int S, int s, int C, int H, int W,
                                 float* __restrict__ dx) {
  const int64_t total = (int64_t)N * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / (H * W));
    const int yx = (int)(i - (int64_t)n * H * W);
    const int y = yx / W, x = yx - y * W;
    for (int c0 = 0; c0 < C; c0 += 4) {
      const float4 v = fold_read(dxpad, ldp, n, y, x, H, W, c0);
      const float vv[4] = {v.x, v.y, v.z, v.w};
      for (int j = 0; j < 4 && c0 + j < C; ++j) dx[(((int64_t)n * S + s) * C + c0 + j) * H * W + yx] = vv[j];
    }
  }
}

int unpack_dx_launch(const void* dxpad, int dt, int ldp, int N, int S, int s, int C, int H, int W, float* dx, hipStream_t st) {
  const int64_t total = (int64_t)N * H * W;
  const int blocks = (int)std::min<int64_t>(ceil_div64(total, 256), 4096);
  MIMO_ST_DISPATCH(dt, T, hipLaunchKernelGGL(unpack_dx_kernel<T>, dim3(blocks), dim3(256), 0, st, (const T*)dxpad, ldp, N, S, s,
                                             C, H, W, dx));
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// ---------------------------------------------------------------------------------------
// BatchNorm statistics -> scale/shift
// ---------------------------------------------------------------------------------------
__global__ void bn_fwd_finalize_kernel(const double* __restrict__ sums, int chunks, int cout_pad, int C, int Cp,
                                       double count, const float* __restrict__ gamma, const float* __restrict__ beta,
                                       float* __restrict__ running_mean, float* __restrict__ running_var,
                                       float momentum, float eps, float* __restrict__ mean, float* __restrict__ invstd,
                                       float* __restrict__ scale, float* __restrict__ shift) {
  __shared__ double red[512];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int cols = 2 * cout_pad;
  double s1, s2;
  chunk_total2(sums + c, sums + cout_pad + c, cols, chunks, c < C, red, &s1, &s2);
  if (c >= Cp || threadIdx.x >= 64) return;
  if (c >= C) {
    mean[c] = 0.f;
    invstd[c] = 0.f;
    scale[c] = 0.f;
    shift[c] = 0.f;
    return;
  }
  const double m = s1 / count;
  double var = s2 / count - m * m;
  var = var > 0.0 ? var : 0.0;
  const double is = 1.0 / sqrt(var + (double)eps);
  const double sc = (double)gamma[c] * is;
  mean[c] = (float)m;
  invstd[c] = (float)is;
  scale[c] = (float)sc;
  shift[c] = (float)((double)beta[c] - m * sc);
  const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
  running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * m);
  running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
}

// rowsum + bn_fwd_finalize in one launch: partial rows [rows][2][cout_pad] straight from the convolution epilogue
__global__ __launch_bounds__(kColsumThreads) void bn_fwd_stats_kernel(
    const float* __restrict__ partial, int rows, int cout_pad, int C, int Cp, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
    float eps, float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift,
    double* __restrict__ scratch, int scratch_cols, int* __restrict__ tickets, int* __restrict__ status) {
  __shared__ double red[2048];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  double s1, s2;
  if (!grid_colsum2(partial, rows, (size_t)2 * cout_pad, c, cout_pad + c, c < C, true, red, scratch, scratch_cols, tickets, &s1,
                    &s2))
    return;
  if (c >= Cp || threadIdx.x >= 64) return;
  if (c >= C) {
    mean[c] = 0.f;
    invstd[c] = 0.f;
    scale[c] = 0.f;
    shift[c] = 0.f;
    return;
  }
  // a convolution output that is not finite (an input / weight outside the fp16 range of the split forward, a
  // diverged run) shows in its channel sums: recorded for mimo_plan_status instead of surfacing only as NaNs downstream
  if (status && !(isfinite(s1) && isfinite(s2))) atomicOr(status, kStatusFwdStats);
  const double m = s1 / count;
  double var = s2 / count - m * m;
  var = var > 0.0 ? var : 0.0;
  const double is = 1.0 / sqrt(var + (double)eps);
  const double sc = (double)gamma[c] * is;
  mean[c] = (float)m;
  invstd[c] = (float)is;
  scale[c] = (float)sc;
  shift[c] = (float)((double)beta[c] - m * sc);
  const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
  running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * m);
  running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
}

int bn_fwd_stats_launch(const float* partial, int rows, int cout_pad, int C, int Cp, int64_t count, const float* gamma,
                        const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* mean,
                        float* invstd, float* scale, float* shift, const ColsumScratch& cs, hipStream_t st) {
  const int groups = ceil_div(Cp, 64);
  hipLaunchKernelGGL(bn_fwd_stats_kernel, dim3(groups, colsum_chunks(rows)), dim3(kColsumThreads), 0, st, partial, rows,
                     cout_pad, C, Cp, (double)count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale,
                     shift, cs.sums, groups * 64, cs.tickets, cs.status);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

int bn_fwd_finalize_launch(const double* sums, int chunks, int cout_pad, int C, int Cp, int64_t count,
                           const float* gamma, const float* beta, float* running_mean, float* running_var,
                           float momentum, float eps, float* mean, float* invstd, float* scale, float* shift,
                           hipStream_t st) {
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(ceil_div(Cp, 64)), dim3(256), 0, st, sums, chunks, cout_pad, C, Cp,
                     (double)count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale, shift);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

__global__ void bn_eval_prepare_kernel(int C, int Cp, const float* __restrict__ gamma, const float* __restrict__ beta,
                                       const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                       float eps, float* __restrict__ mean, float* __restrict__ invstd,
                                       float* __restrict__ scale, float* __restrict__ shift, int* __restrict__ status) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= Cp) return;
  if (c >= C) {
    mean[c] = 0.f;
    invstd[c] = 0.f;
    scale[c] = 0.f;
    shift[c] = 0.f;
    return;
  }
  const double m = running_mean[c];
  const double is = 1.0 / sqrt((double)running_var[c] + (double)eps);
  const double sc = (double)gamma[c] * is;
  // running statistics poisoned by an earlier training step (a NaN shift would be dropped by the ReLU's fmaxf: a silently
  // dead channel in eval mode)
  if (status && !(isfinite(m) && isfinite(is) && isfinite(sc))) atomicOr(status, kStatusFwdStats);
  mean[c] = (float)m;
  invstd[c] = (float)is;
  scale[c] = (float)sc;
  shift[c] = (float)((double)beta[c] - m * sc);
}

int bn_eval_prepare_launch(int C, int Cp, const float* gamma, const float* beta, const float* running_mean,
                           const float* running_var, float eps, float* mean, float* invstd, float* scale,
                           float* shift, hipStream_t st, int* status) {
  hipLaunchKernelGGL(bn_eval_prepare_kernel, dim3(ceil_div(Cp, 64)), dim3(64), 0, st, C, Cp, gamma, beta,
                     running_mean, running_var, eps, mean, invstd, scale, shift, status);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

__device__ __forceinline__ float4 mask4(const float* mask, int n, int C, int c0) {
  float4 m;
  const float* mp = mask + (size_t)n * C;
  m.x = c0 + 0 < C ? mp[c0 + 0] : 0.f;
  m.y = c0 + 1 < C ? mp[c0 + 1] : 0.f;
  m.z = c0 + 2 < C ? mp[c0 + 2] : 0.f;
  m.w = c0 + 3 < C ? mp[c0 + 3] : 0.f;
  return m;
}

template <typename TZ, typename TA>
__global__ void bn_relu_fwd_kernel(const TZ* __restrict__ z, int ldz, TA* __restrict__ a, int lda,
                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                   const float* __restrict__ mask, int C, int Cv, int P, int HW, int* __restrict__ status) {
  const PQ t = pixquad(Cv);
  if (!t.active) return;
  const float4 sc = ld4(scale + 4 * t.q), sh = ld4(shift + 4 * t.q);
  for (int p = t.p; p < P; p += t.pstep) {
    const float4 v = ld4(z + (size_t)p * ldz + 4 * t.q);
    // eval mode (status != nullptr; a training forward sees it in the batch statistics): fmaxf drops a NaN, so a
    // convolution output that is not finite would otherwise vanish here as a silently dead channel
    if (status && !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w))) atomicOr(status, kStatusFwdStats);
    float4 r;
    r.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f);
    r.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
    r.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f);
    r.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
    if (mask) {
      const float4 m = mask4(mask, p / HW, C, 4 * t.q);
      r.x *= m.x;
      r.y *= m.y;
      r.z *= m.z;
      r.w *= m.w;
    }
    st4(a + (size_t)p * lda + 4 * t.q, r);
  }
}

int bn_relu_fwd_launch(const void* z, int dtz, int ldz, void* a, int dta, int lda, const float* scale, const float* shift,
                       const float* mask, int C, int Cp, int64_t P, int HW, hipStream_t st, int* status) {
  const int Cv = Cp / 4;
  MIMO_ST_DISPATCH2(dtz, dta, TZ, TA,
                    hipLaunchKernelGGL((bn_relu_fwd_kernel<TZ, TA>), pq_grid(Cv, P, kBlocksBnRelu), dim3(256), 0, st,
                                       (const TZ*)z, ldz, (TA*)a, lda, scale, shift, mask, C, Cv, (int)P, HW, status))
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// BatchNorm + ReLU (+ Dropout2d multipliers) of a tensor whose only spatial consumer is MaxPool2d(2): one thread
// owns a 2x2 window, writes its four activations and their maximum into the pooled tensor — the separate
// pooling pass (a second read of the activation) disappears.  Same arithmetic, bit-identical results.
template <typename TZ, typename TA>
__global__ void bn_relu_pool_fwd_kernel(const TZ* __restrict__ z, int ldz, TA* __restrict__ a, int lda,
                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                        const float* __restrict__ mask, int C, int Cv, int N, int H, int W,
                                        TA* __restrict__ pool, int ldpool, int* __restrict__ status) {
  const PQ t = pixquad(Cv);
  if (!t.active) return;
  const float4 sc = ld4(scale + 4 * t.q), sh = ld4(shift + 4 * t.q);
  const int Hp = H / 2, Wp = W / 2;
  const int P = N * Hp * Wp;
  auto act = [&](float4 v, float4 m) {
    if (status && !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w))) atomicOr(status, kStatusFwdStats);  // as bn_relu_fwd_kernel
    float4 r;
    r.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f) * m.x;
    r.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f) * m.y;
    r.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f) * m.z;
    r.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f) * m.w;
    return r;
  };
  PixIter it = pix_iter(t.p, t.pstep, Hp, Wp);
  for (int p = t.p; p < P; p += t.pstep, pix_next(it, Hp, Wp)) {
    const int n = it.n, py = it.y, px = it.x;
    const float4 m = mask ? mask4(mask, n, C, 4 * t.q) : make_float4(1.f, 1.f, 1.f, 1.f);
    const size_t pix = ((size_t)n * H + 2 * py) * W + 2 * px;
    const TZ* zs = z + pix * ldz + 4 * t.q;
    TA* as = a + pix * lda + 4 * t.q;
    const float4 z00 = ld4(zs), z01 = ld4(zs + ldz), z10 = ld4(zs + (size_t)W * ldz), z11 = ld4(zs + (size_t)(W + 1) * ldz);
    const float4 r00 = act(z00, m), r01 = act(z01, m), r10 = act(z10, m), r11 = act(z11, m);
    st4(as, r00);
    st4(as + lda, r01);
    st4(as + (size_t)W * lda, r10);
    st4(as + (size_t)(W + 1) * lda, r11);
    // rounding to the storage type is monotonic: max of the rounded values == rounded max
    st4(pool + (size_t)p * ldpool + 4 * t.q, f4max(f4max(r00, r01), f4max(r10, r11)));
    // odd sizes: the last column / row belongs to no window but is still an activation
    if ((W & 1) && px == Wp - 1) {
      st4(as + 2 * (size_t)lda, act(ld4(zs + 2 * (size_t)ldz), m));
      st4(as + (size_t)(W + 2) * lda, act(ld4(zs + (size_t)(W + 2) * ldz), m));
    }
    if ((H & 1) && py == Hp - 1) {
      st4(as + 2 * (size_t)W * lda, act(ld4(zs + 2 * (size_t)W * ldz), m));
      st4(as + (2 * (size_t)W + 1) * lda, act(ld4(zs + (2 * (size_t)W + 1) * ldz), m));
      if ((W & 1) && px == Wp - 1) st4(as + (2 * (size_t)W + 2) * lda, act(ld4(zs + (2 * (size_t)W + 2) * ldz), m));
    }
  }
}

int bn_relu_pool_fwd_launch(const void* z, int dtz, int ldz, void* a, int dta, int lda, const float* scale, const float* shift,
                            const float* mask, int C, int Cp, int N, int H, int W, void* pool, int ldpool, hipStream_t st,
                            int* status) {
  const int Cv = Cp / 4;
  MIMO_ST_DISPATCH2(dtz, dta, TZ, TA,
                    hipLaunchKernelGGL((bn_relu_pool_fwd_kernel<TZ, TA>), pq_grid(Cv, (int64_t)N * (H / 2) * (W / 2), 4096),
                                       dim3(256), 0, st, (const TZ*)z, ldz, (TA*)a, lda, scale, shift, mask, C, Cv, N, H, W,
                                       (TA*)pool, ldpool, status))
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// ---------------------------------------------------------------------------------------
// max pooling 2x2 (floor) and bilinear x2 upsample + zero pad + concat
// ---------------------------------------------------------------------------------------

template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ a, int lda, int N, int H, int W, int Cv,
                                   T* __restrict__ out, int ldo) {
  const PQ t = pixquad(Cv);
  if (!t.active) return;
  const int Ho = H / 2, Wo = W / 2;
  const int P = N * Ho * Wo;
  PixIter it = pix_iter(t.p, t.pstep, Ho, Wo);
  for (int p = t.p; p < P; p += t.pstep, pix_next(it, Ho, Wo)) {
    const int n = it.n, oy = it.y, ox = it.x;
    const T* src = a + (((size_t)n * H + 2 * oy) * W + 2 * ox) * lda + 4 * t.q;
    const float4 v = f4max(f4max(ld4(src), ld4(src + lda)), f4max(ld4(src + (size_t)W * lda), ld4(src + (size_t)(W + 1) * lda)));
    st4(out + (size_t)p * ldo + 4 * t.q, v);
  }
}

int maxpool_fwd_launch(const void* a, int dt, int lda, int N, int H, int W, int Cp, void* out, int ldo, hipStream_t st) {
  const int Cv = Cp / 4;
  const int64_t P = (int64_t)N * (H / 2) * (W / 2);
  MIMO_ST_DISPATCH(dt, T, hipLaunchKernelGGL(maxpool_fwd_kernel<T>, pq_grid(Cv, P, 4096), dim3(256), 0, st, (const T*)a, lda, N, H,
                                             W, Cv, (T*)out, ldo));
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// source index / weight of bilinear align_corners=True interpolation, as torch computes it in
// fp32: src = dst * (in-1)/(out-1); i0 = (int)src; lambda = src - i0; i1 = i0 + (i0 < in-1)
struct Lerp {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ Lerp lerp_src(int dst, int in, int out) {
  // no fma contraction in here: torch rounds src = scale * dst to fp32 and then subtracts i0 (checked
  // against F.interpolate on the CPU); an fma keeps the exact product and moves lambda by up to an ulp of
  // src (4e-6 at src ~ 50, 2e-6 relative on the output) — and whether the compiler contracts depends on
  // the surrounding code, so it is pinned here
#pragma clang fp contract(off)
  const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  const float src = scale * (float)dst;
  Lerp r;
  r.i0 = min((int)src, in - 1);
  r.l1 = fminf(fmaxf(src - (float)r.i0, 0.f), 1.f);
  r.l0 = 1.f - r.l1;
  r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
  return r;
}

// weight with which output index o (of an x2 align_corners upsample of `in` samples) reads input i
__device__ __forceinline__ float lerp_weight(int o, int i, int in) {
  const Lerp l = lerp_src(o, in, 2 * in);
  return (l.i0 == i ? l.l0 : 0.f) + (l.i1 == i ? l.l1 : 0.f);
}

// lo_scale != nullptr: `low` is the pre-activation tensor of the producing convolution and its BatchNorm + ReLU is applied to
// the four sources of every output pixel here (relu(fma(z, scale, shift)), bn_relu_fwd_kernel's arithmetic, then the same
// interpolation): the activated low-resolution tensor is never written (round 4)
__device__ __forceinline__ float4 bn_relu4(float4 v, float4 sc, float4 sh) {
  return make_float4(fmaxf(fmaf(v.x, sc.x, sh.x), 0.f), fmaxf(fmaf(v.y, sc.y, sh.y), 0.f), fmaxf(fmaf(v.z, sc.z, sh.z), 0.f),
                     fmaxf(fmaf(v.w, sc.w, sh.w), 0.f));
}

__device__ __forceinline__ float4 sel3(int a, float4 x0, float4 x1, float4 x2) { return a == 1 ? x1 : (a == 0 ? x0 : x2); }
__device__ __forceinline__ float4 blend2(float w0, float4 a, float w1, float4 b) {
  // no fma contraction (here and in the per-pixel kernel's blend): both kernels then round every product and sum alike
#pragma clang fp contract(off)
  return make_float4(w0 * a.x + w1 * b.x, w0 * a.y + w1 * b.y, w0 * a.z + w1 * b.z, w0 * a.w + w1 * b.w);
}
template <typename T>
__global__ void upcat_fwd_kernel(const T* __restrict__ skip, int lds, int csv, const T* __restrict__ low,
                                 int ldl, int clv, int N, int H, int W, int h, int w, int padT, int padL,
                                 T* __restrict__ out, const float* __restrict__ lo_scale,
                                 const float* __restrict__ lo_shift) {
  const int Cv = csv + clv;
  // skip == nullptr: the skip tensor already lives in channels [0, 4*csv) of `out` (its producer writes it
  // there); only the up-sampled part is written, and the threads are mapped over those channels alone
  PQ t = pixquad(skip ? Cv : clv);
  if (!t.active) return;
  if (!skip) t.q += csv;
  const int ldo = 4 * Cv;
  const int P = N * H * W;
  float4 lsc = f4zero(), lsh = f4zero();
  if (lo_scale && t.q >= csv) {
    lsc = ld4(lo_scale + 4 * (t.q - csv));
    lsh = ld4(lo_shift + 4 * (t.q - csv));
  }
  PixIter it = pix_iter(t.p, t.pstep, H, W);
  for (int p = t.p; p < P; p += t.pstep, pix_next(it, H, W)) {
    float4 v;
    if (t.q < csv) {
      v = ld4(skip + (size_t)p * lds + 4 * t.q);
    } else {
      const int n = it.n, y = it.y, x = it.x;
      const int uy = y - padT, ux = x - padL;
      v = f4zero();
      if (uy >= 0 && uy < 2 * h && ux >= 0 && ux < 2 * w) {
        const Lerp ly = lerp_src(uy, h, 2 * h), lx = lerp_src(ux, w, 2 * w);
        const T* b = low + (size_t)n * h * w * ldl + 4 * (t.q - csv);
        float4 v00 = ld4(b + ((size_t)ly.i0 * w + lx.i0) * ldl), v01 = ld4(b + ((size_t)ly.i0 * w + lx.i1) * ldl);
        float4 v10 = ld4(b + ((size_t)ly.i1 * w + lx.i0) * ldl), v11 = ld4(b + ((size_t)ly.i1 * w + lx.i1) * ldl);
        if (lo_scale) {
          v00 = bn_relu4(v00, lsc, lsh);
          v01 = bn_relu4(v01, lsc, lsh);
          v10 = bn_relu4(v10, lsc, lsh);
          v11 = bn_relu4(v11, lsc, lsh);
        }
        v = blend2(ly.l0, blend2(lx.l0, v00, lx.l1, v01), ly.l1, blend2(lx.l0, v10, lx.l1, v11));
      }
    }
    st4(out + (size_t)p * ldo + 4 * t.q, v);
  }
}

// The common geometry — skip tensor already in place (skip == nullptr), output exactly twice the low-resolution size — with
// one thread per 2 x 2 OUTPUT block and channel quad (round 5): the four pixels of a block read sources
// from the 3 x 3 low-resolution neighbourhood of their block only (x2 align_corners: output 2k reads rows k-1 / k, output
// 2k + 1 rows k / k + 1), so 9 loads (and 9 BatchNorm + ReLU evaluations) serve 4 outputs instead of 16.  Which of the three
// rows / columns an output takes is read off the same lerp_src() results as in the per-pixel kernel, and the blend keeps its
// association — horizontal pairs first, then the vertical pair: the same values.
template <typename T>
__global__ __launch_bounds__(256) void upcat_fwd2x2_kernel(int csv, const T* __restrict__ low, int ldl, int clv, int N, int h, int w,
                                    T* __restrict__ out, const float* __restrict__ lo_scale,
                                    const float* __restrict__ lo_shift) {
  PQ t = pixquad(clv);
  if (!t.active) return;
  const int ldo = 4 * (csv + clv), H = 2 * h, W = 2 * w;
  const int P = N * h * w;
  float4 lsc = f4zero(), lsh = f4zero();
  if (lo_scale) {
    lsc = ld4(lo_scale + 4 * t.q);
    lsh = ld4(lo_shift + 4 * t.q);
  }
  PixIter it = pix_iter(t.p, t.pstep, h, w);
  for (int p = t.p; p < P; p += t.pstep, pix_next(it, h, w)) {
    const int n = it.n, by = it.y, bx = it.x;
    const int rr[3] = {max(by - 1, 0), by, min(by + 1, h - 1)}, cc[3] = {max(bx - 1, 0), bx, min(bx + 1, w - 1)};
    const T* b = low + (size_t)n * h * w * ldl + 4 * t.q;
    float4 v[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) v[a][c] = ld4(b + ((size_t)rr[a] * w + cc[c]) * ldl);
    if (lo_scale) {
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) v[a][c] = bn_relu4(v[a][c], lsc, lsh);
    }
    const Lerp ly[2] = {lerp_src(2 * by, h, H), lerp_src(2 * by + 1, h, H)};
    const Lerp lx[2] = {lerp_src(2 * bx, w, W), lerp_src(2 * bx + 1, w, W)};
    // horizontal pairs of every source row, for both output columns
    float4 hx[3][2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int j0 = lx[e].i0 - (bx - 1), j1 = lx[e].i1 - (bx - 1);
#pragma unroll
      for (int a = 0; a < 3; ++a)
        hx[a][e] = blend2(lx[e].l0, sel3(j0, v[a][0], v[a][1], v[a][2]), lx[e].l1, sel3(j1, v[a][0], v[a][1], v[a][2]));
    }
    T* o = out + ((size_t)(n * H + 2 * by) * W + 2 * bx) * ldo + 4 * (csv + t.q);
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int a0 = ly[d].i0 - (by - 1), a1 = ly[d].i1 - (by - 1);
#pragma unroll
      for (int e = 0; e < 2; ++e)
        st4(o + ((size_t)d * W + e) * ldo,
            blend2(ly[d].l0, sel3(a0, hx[0][e], hx[1][e], hx[2][e]), ly[d].l1, sel3(a1, hx[0][e], hx[1][e], hx[2][e])));
    }
  }
}

int upcat_fwd_launch(const void* skip, int dt, int lds, int csp, const void* low, int ldl, int clp, int N, int H, int W,
                     int h, int w, void* out, hipStream_t st, const float* lo_scale, const float* lo_shift) {
  const int Cv = (csp + clp) / 4;
  const int padT = (H - 2 * h) / 2, padL = (W - 2 * w) / 2;  // F.pad(diff//2, diff - diff//2), components.py:110-115
  if (H < 2 * h || W < 2 * w) {
    set_error("upcat: skip smaller than upsampled input");
    return MIMO_ERR_INVALID;
  }
  static const bool blocks2x2 = !(getenv("MIMO_UPCAT_2X2") && atoi(getenv("MIMO_UPCAT_2X2")) == 0);
  if (!skip && H == 2 * h && W == 2 * w && blocks2x2) {
    MIMO_ST_DISPATCH(dt, T, hipLaunchKernelGGL(upcat_fwd2x2_kernel<T>, pq_grid(clp / 4, (int64_t)N * h * w, 4096), dim3(256), 0, st,
                                               csp / 4, (const T*)low, ldl, clp / 4, N, h, w, (T*)out, lo_scale, lo_shift));
    MIMO_KERNEL_CHECK();
    return MIMO_OK;
  }
  MIMO_ST_DISPATCH(dt, T, hipLaunchKernelGGL(upcat_fwd_kernel<T>, pq_grid(skip ? Cv : clp / 4, (int64_t)N * H * W, 4096), dim3(256), 0,
                                             st, (const T*)skip, lds, csp / 4, (const T*)low, ldl, clp / 4, N, H, W, h, w, padT, padL,
                                             (T*)out, lo_scale, lo_shift));
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// ---------------------------------------------------------------------------------------
// backward gathers
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float route_max(float g, float v00, float v01, float v10, float v11, int self) {
  // first maximum in scan order (0,0),(0,1),(1,0),(1,1) receives the gradient (strict '>' update)
  int arg = 0;
  float m = v00;
  if (v01 > m) { m = v01; arg = 1; }
  if (v10 > m) { m = v10; arg = 2; }
  if (v11 > m) { m = v11; arg = 3; }
  return arg == self ? g : 0.f;
}

// One thread owns a 2x2 window: one folded gradient read, the four activations once (not once per
// window member), four stores.  Rows / columns outside every window (odd H or W) receive no pooled gradient.
// skip != nullptr: the tensor is also the skip input of an Up block; that block's padded-domain data gradient
// (channels [0, Cp) of its own buffer, pixel pitch ldsk) is folded and added here, so the skip slice is never
// copied out (da = pool route + skip fold, the same two operands in the same order as the former
// fold_slice + accumulating pool_bwd pair).
template <typename T>
__global__ void pool_bwd_kernel(const T* __restrict__ dxpad, int ldp, int choff, const T* __restrict__ a,
                                int lda, T* __restrict__ da, int ldda, int N, int H, int W, int Cv, int accumulate,
                                const T* __restrict__ skip, int ldsk) {
  const PQ t = pixquad(Cv);
  if (!t.active) return;
  const int Hp = H / 2, Wp = W / 2;
  const int P = N * Hp * Wp;
  PixIter it = pix_iter(t.p, t.pstep, Hp, Wp);
  for (int p = t.p; p < P; p += t.pstep, pix_next(it, Hp, Wp)) {
    const int n = it.n, py = it.y, px = it.x;
    const float4 g = fold_read(dxpad, ldp, n, py, px, Hp, Wp, choff + 4 * t.q);
    const size_t pix = ((size_t)n * H + 2 * py) * W + 2 * px;
    const T* src = a + pix * lda + 4 * t.q;
    const float4 v00 = ld4(src), v01 = ld4(src + lda), v10 = ld4(src + (size_t)W * lda), v11 = ld4(src + (size_t)(W + 1) * lda);
    T* dst = da + pix * ldda + 4 * t.q;
    float4 r[4];
#pragma unroll
    for (int self = 0; self < 4; ++self) {
      r[self].x = route_max(g.x, v00.x, v01.x, v10.x, v11.x, self);
      r[self].y = route_max(g.y, v00.y, v01.y, v10.y, v11.y, self);
      r[self].z = route_max(g.z, v00.z, v01.z, v10.z, v11.z, self);
      r[self].w = route_max(g.w, v00.w, v01.w, v10.w, v11.w, self);
    }
    T* d4[4] = {dst, dst + ldda, dst + (size_t)W * ldda, dst + (size_t)(W + 1) * ldda};
    if (accumulate) {
      float4 o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = ld4(d4[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = f4add(r[j], o[j]);
    }
    if (skip) {
      float4 o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = fold_read(skip, ldsk, n, 2 * py + (j >> 1), 2 * px + (j & 1), H, W, 4 * t.q);
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = f4add(r[j], o[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) st4(d4[j], r[j]);
    if (!accumulate) {  // odd sizes: the last row / column belongs to no window (skip gradient only)
      auto rest = [&](int y, int x) {
        st4(da + (((size_t)n * H + y) * W + x) * ldda + 4 * t.q, skip ? fold_read(skip, ldsk, n, y, x, H, W, 4 * t.q) : f4zero());
      };
      if ((W & 1) && px == Wp - 1) {
        rest(2 * py, W - 1);
        rest(2 * py + 1, W - 1);
      }
      if ((H & 1) && py == Hp - 1) {
        rest(H - 1, 2 * px);
        rest(H - 1, 2 * px + 1);
        if ((W & 1) && px == Wp - 1) rest(H - 1, W - 1);
      }
    } else if (skip) {
      auto rest = [&](int y, int x) {
        T* d = da + (((size_t)n * H + y) * W + x) * ldda + 4 * t.q;
        st4(d, f4add(ld4(d), fold_read(skip, ldsk, n, y, x, H, W, 4 * t.q)));
      };
      if ((W & 1) && px == Wp - 1) {
        rest(2 * py, W - 1);
        rest(2 * py + 1, W - 1);
      }
      if ((H & 1) && py == Hp - 1) {
        rest(H - 1, 2 * px);
        rest(H - 1, 2 * px + 1);
        if ((W & 1) && px == Wp - 1) rest(H - 1, W - 1);
      }
    }
  }
}

int pool_bwd_launch(const void* dxpad, int dt, int ldp, int choff, const void* a, int lda, void* da, int ldda, int N, int H,
                    int W, int Cp, int accumulate, hipStream_t st, const void* skip, int ldsk) {
  const int Cv = Cp / 4;
  MIMO_ST_DISPATCH(dt, T, hipLaunchKernelGGL(pool_bwd_kernel<T>, pq_grid(Cv, (int64_t)N * (H / 2) * (W / 2), kBlocksPoolBwd), dim3(256),
                                             0, st, (const T*)dxpad, ldp, choff, (const T*)a, lda, (T*)da, ldda, N, H, W, Cv,
                                             accumulate, (const T*)skip, ldsk));
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

template <typename T>
__global__ void fold_slice_kernel(const T* __restrict__ dxpad, int ldp, int choff, T* __restrict__ da, int ldda,
                                  int N, int H, int W, int Cv, int accumulate) {
  const PQ t = pixquad(Cv);
  if (!t.active) return;
  const int P = N * H * W;
  PixIter it = pix_iter(t.p, t.pstep, H, W);
  for (int p = t.p; p < P; p += t.pstep, pix_next(it, H, W)) {
    float4 v = fold_read(dxpad, ldp, it.n, it.y, it.x, H, W, choff + 4 * t.q);
    T* dst = da + (size_t)p * ldda + 4 * t.q;
    if (accumulate) v = f4add(v, ld4(dst));
    st4(dst, v);
  }
}

int fold_slice_launch(const void* dxpad, int dt, int ldp, int choff, void* da, int ldda, int N, int H, int W, int Cp,
                      int accumulate, hipStream_t st) {
  const int Cv = Cp / 4;
  MIMO_ST_DISPATCH(dt, T, hipLaunchKernelGGL(fold_slice_kernel<T>, pq_grid(Cv, (int64_t)N * H * W, 4096), dim3(256), 0, st,
                                             (const T*)dxpad, ldp, choff, (T*)da, ldda, N, H, W, Cv, accumulate));
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// ---- in-engine dropout masks: Philox4x32-10 (Salmon et al., the counter-based generator torch's CUDA / HIP dropout
// uses), keyed by a (seed, offset) pair the caller takes from its generator.  A multiplier depends only on (seed,
// offset, site, element index), so the backward regenerates exactly what the forward used and nothing is stored for
// the element-wise sites.  Statistically equivalent to the reference's Bernoulli draws (keep with probability 1 - p,
// scale by 1 / (1 - p)); bit-level parity with recorded masks stays available through the mask arguments.
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
  constexpr unsigned int M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned int hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
    const unsigned int hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
    ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
    key.x += W0;
    key.y += W1;
  }
  return ctr;
}
__device__ __forceinline__ float4 philox_keep4(unsigned int i0, unsigned int i1, unsigned int site, uint64_t seed, uint64_t offset,
                                               float p, float inv_keep) {
  const uint4 r = philox4x32_10(make_uint4(i0, i1, (unsigned int)offset ^ (site << 24), (unsigned int)(offset >> 32)),
                                make_uint2((unsigned int)seed, (unsigned int)(seed >> 32)));
  constexpr float k = 1.f / 16777216.f;  // 24 uniform bits
  return make_float4((r.x >> 8) * k >= p ? inv_keep : 0.f, (r.y >> 8) * k >= p ? inv_keep : 0.f,
                     (r.z >> 8) * k >= p ? inv_keep : 0.f, (r.w >> 8) * k >= p ? inv_keep : 0.f);
}

// One wave that does nothing for `us` microseconds (s_memrealtime: the constant 100 MHz counter).  Test hook only
// (MIMO_DEBUG_WGRAD_DELAY_US, plan.hip): delays the side stream in front of every weight gradient so that the stream
// protocol's hazards — a buffer released before its last reader has run — show as wrong bits instead of staying hidden
// behind favourable timing.
__global__ void debug_delay_kernel(int us, int* __restrict__ sink) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  const uint64_t ticks = (uint64_t)us * 100u;
  int n = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    __builtin_amdgcn_s_sleep(32);
    ++n;
  }
  if (sink && n < 0) *sink = n;  // never taken: keeps the loop
}
int debug_delay_launch(int us, hipStream_t st) {
  hipLaunchKernelGGL(debug_delay_kernel, dim3(1), dim3(64), 0, st, us, (int*)nullptr);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// Dropout2d multipliers [N][C] of every active site in one launch (blockIdx.y = site, bit `site` of `active`)
__global__ void dropout2d_masks_kernel(const Dropout2dSite* __restrict__ sites, uint64_t active, uint64_t seed, uint64_t offset) {
  const int site = blockIdx.y;
  if (!((active >> site) & 1)) return;
  const Dropout2dSite t = sites[site];
  const float inv_keep = 1.f / (1.f - t.p);
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; 4 * q < t.count; q += gridDim.x * blockDim.x) {
    const float4 m = philox_keep4((unsigned int)q, 0u, (unsigned int)site, seed, offset, t.p, inv_keep);
    const float mv[4] = {m.x, m.y, m.z, m.w};
    for (int j = 0; j < 4 && 4 * q + j < t.count; ++j) t.dst[4 * q + j] = mv[j];
  }
}

int dropout2d_masks_launch(const Dropout2dSite* sites_dev, int nsites, int max_count, uint64_t active, uint64_t seed,
                           uint64_t offset, hipStream_t st) {
  if (!active || nsites <= 0) return MIMO_OK;
  const int gx = std::max(1, std::min(ceil_div(ceil_div(max_count, 4), 256), 64));
  hipLaunchKernelGGL(dropout2d_masks_kernel, dim3(gx, nsites), dim3(256), 0, st, sites_dev, active, seed, offset);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// the element-wise multipliers of one site in the reference's NCHW layout [N][C][HW] (tests / recorded-mask parity)
__global__ void elem_dropout_mask_kernel(float* __restrict__ mask, int N, int C, int Cv, int HW, unsigned int site, uint64_t seed,
                                         uint64_t offset, float p) {
  const float inv_keep = 1.f / (1.f - p);
  const int64_t total = (int64_t)N * Cv * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i % HW);
    const int nq = (int)(i / HW);
    const int n = nq / Cv, q = nq - n * Cv;
    const float4 m = philox_keep4((unsigned int)r, (unsigned int)nq, site, seed, offset, p, inv_keep);
    const float mv[4] = {m.x, m.y, m.z, m.w};
    for (int j = 0; j < 4 && 4 * q + j < C; ++j) mask[((size_t)n * C + 4 * q + j) * HW + r] = mv[j];
  }
}

int elem_dropout_mask_launch(float* mask, int N, int C, int Cp, int HW, int site, uint64_t seed, uint64_t offset, float p,
                             hipStream_t st) {
  const int64_t total = (int64_t)N * (Cp / 4) * HW;
  const int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(elem_dropout_mask_kernel, dim3(blocks), dim3(256), 0, st, mask, N, C, Cp / 4, HW, (unsigned int)site, seed,
                     offset, p);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// ---- element-wise dropout (nn.Dropout after down4 and in front of each 1x1 head) -------------
// a[n,p,c] *= mask[n,c,p]: the multipliers arrive in the reference's NCHW layout, so one thread owns a
// pixel (coalesced mask reads per channel plane) and walks its own contiguous channel row of `a`.
// mask == nullptr: the multipliers come from the Philox stream (rng), regenerated identically by the backward
template <typename T>
__global__ void elem_mask_mul_kernel(T* __restrict__ a, int ld, const float* __restrict__ mask, int N, int C, int Cv,
                                     int HW, ElemRng rng) {
  const int64_t P = (int64_t)N * HW;
  const float inv_keep = 1.f / (1.f - rng.p);
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(p / HW);
    const int r = (int)(p - (int64_t)n * HW);
    const float* m = mask + (size_t)n * C * HW + r;
    T* row = a + (size_t)p * ld;
    for (int q = 0; q < Cv; ++q) {
      float4 v = ld4(row + 4 * q);
      const int c = 4 * q;
      if (mask) {
        v.x *= c + 0 < C ? m[(size_t)(c + 0) * HW] : 1.f;
        v.y *= c + 1 < C ? m[(size_t)(c + 1) * HW] : 1.f;
        v.z *= c + 2 < C ? m[(size_t)(c + 2) * HW] : 1.f;
        v.w *= c + 3 < C ? m[(size_t)(c + 3) * HW] : 1.f;
      } else {
        const float4 k = philox_keep4((unsigned int)r, (unsigned int)(n * Cv + q), (unsigned int)rng.site, rng.seed, rng.offset,
                                      rng.p, inv_keep);
        v.x *= k.x;
        v.y *= k.y;
        v.z *= k.z;
        v.w *= k.w;
      }
      st4(row + 4 * q, v);
    }
  }
}

int elem_mask_mul_launch(void* a, int dt, int ld, const float* mask, int N, int C, int Cp, int HW, hipStream_t st,
                         const ElemRng* rng) {
  const int64_t P = (int64_t)N * HW;
  const int blocks = (int)std::min<int64_t>((P + 255) / 256, 4096);
  if (!mask && !rng) {
    set_error("elem_mask_mul: neither a mask nor a generator state");
    return MIMO_ERR_INVALID;
  }
  const ElemRng g = rng ? *rng : ElemRng{0, 0, 0, 0.f};
  MIMO_ST_DISPATCH(dt, T, hipLaunchKernelGGL(elem_mask_mul_kernel<T>, dim3(blocks), dim3(256), 0, st, (T*)a, ld, mask, N, C, Cp / 4, HW, g));
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// Input pixel i of an x2 align_corners upsample is read by outputs 2i-1 .. 2i+2 only (2i-2 lands on
// i-2 / i-1 for every size; checked exhaustively for in <= 300 and 512..2048 on the CPU).
// Gradient arriving at low-resolution pixel (n, iy, ix), channel quad q: the transposed interpolation over its 4 x 4
// candidates of the padded-domain data gradient, border folds included.
template <typename T>
__device__ __forceinline__ float4 up_bwd_pixel(const T* __restrict__ dxpad, int ldp, int choff, int q, int n, int iy, int ix,
                                               int H, int W, int h, int w, int padT, int padL) {
    float wy[4], wx[4];
    int cy[4], cx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int oy = 2 * iy - 1 + k, ox = 2 * ix - 1 + k;
      wy[k] = (oy >= 0 && oy < 2 * h) ? lerp_weight(oy, iy, h) : 0.f;
      wx[k] = (ox >= 0 && ox < 2 * w) ? lerp_weight(ox, ix, w) : 0.f;
      cy[k] = min(max(oy, 0), 2 * h - 1) + padT;
      cx[k] = min(max(ox, 0), 2 * w - 1) + padL;
    }
    float4 v = f4zero();
    const T* base = dxpad + (size_t)n * (H + 2) * (W + 2) * ldp + choff + 4 * q;
    // the 16 candidates themselves (padded-domain pixel of image pixel (y, x) = (y + 1, x + 1)): 16 independent loads,
    // weights 0 where the candidate does not read this input (adds an exact zero, same sum as skipping it)
    {
      float4 g[4][4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) g[k][j] = ld4(base + ((size_t)(cy[k] + 1) * (W + 2) + (cx[j] + 1)) * ldp);
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float ww = wy[k] * wx[j];
          v.x += ww * g[k][j].x;
          v.y += ww * g[k][j].y;
          v.z += ww * g[k][j].z;
          v.w += ww * g[k][j].w;
        }
    }
    if (!(cy[0] >= 2 && cy[3] <= H - 3 && cx[0] >= 2 && cx[3] <= W - 3)) {
      // Near the border some candidates also receive a reflected pad row / column (transpose of the reflect padding: pad
      // row -1 folds onto row 1, pad row H onto row H - 2; columns alike).  Folding is linear, so instead of folding every
      // candidate (up to 4 dependent loads each, 16 times, in divergent loops: the whole launch waited for these threads —
      // 94 us at 4 images per GPU where the data moves in 15) the pad rows / columns enter as two more rows and columns of
      // the weighted sum, with the weight of the image row / column they fold onto: at most 20 more independent loads.
      float ey[2] = {0.f, 0.f}, ex[2] = {0.f, 0.f};  // weight of pad row -1 / H, pad column -1 / W
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ey[0] += cy[k] == 1 ? wy[k] : 0.f;
        ey[1] += cy[k] == H - 2 ? wy[k] : 0.f;
        ex[0] += cx[k] == 1 ? wx[k] : 0.f;
        ex[1] += cx[k] == W - 2 ? wx[k] : 0.f;
      }
      const int pry[2] = {0, H + 1}, prx[2] = {0, W + 1};  // padded-domain indices of the pad rows / columns
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (ey[e] != 0.f) {  // pad row e against the four candidate columns and the two pad columns
          float4 g[6];
#pragma unroll
          for (int j = 0; j < 4; ++j) g[j] = ld4(base + ((size_t)pry[e] * (W + 2) + (cx[j] + 1)) * ldp);
#pragma unroll
          for (int j = 0; j < 2; ++j) g[4 + j] = ld4(base + ((size_t)pry[e] * (W + 2) + prx[j]) * ldp);
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            const float ww = ey[e] * (j < 4 ? wx[j] : ex[j - 4]);
            v.x += ww * g[j].x;
            v.y += ww * g[j].y;
            v.z += ww * g[j].z;
            v.w += ww * g[j].w;
          }
        }
        if (ex[e] != 0.f) {  // pad column e against the four candidate rows
          float4 g[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) g[k] = ld4(base + ((size_t)(cy[k] + 1) * (W + 2) + prx[e]) * ldp);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float ww = wy[k] * ex[e];
            v.x += ww * g[k].x;
            v.y += ww * g[k].y;
            v.z += ww * g[k].z;
            v.w += ww * g[k].w;
          }
        }
      }
    }
    return v;
}

template <typename T>
__global__ void up_bwd_kernel(const T* __restrict__ dxpad, int ldp, int choff, T* __restrict__ da, int ldda,
                              int N, int H, int W, int h, int w, int padT, int padL, int Cv, int accumulate) {
  const PQ t = pixquad(Cv, true);
  if (!t.active) return;
  const int P = N * h * w;
  PixIter it = pix_iter(t.p, t.pstep, h, w);
  for (int p = t.p; p < P; p += t.pstep, pix_next(it, h, w)) {
    float4 v = up_bwd_pixel(dxpad, ldp, choff, t.q, it.n, it.y, it.x, H, W, h, w, padT, padL);
    T* dst = da + (size_t)p * ldda + 4 * t.q;
    if (accumulate) v = f4add(v, ld4(dst));
    st4(dst, v);
  }
}

// (Round 5 measured the transposed counterpart of upcat_fwd2x2 — one thread per 2 x 2 block of low-resolution pixels, its
// four 4 x 4 candidate windows read as one 6 x 6 window, 36 loads per 4 outputs instead of 64: 0.49 -> 0.52 ms per step,
// slower at 165 registers and three waves per SIMD; profiles/r05/upsampling_2x2.txt.  Not kept.)

int up_bwd_launch(const void* dxpad, int dt, int ldp, int choff, void* da, int ldda, int N, int H, int W, int h, int w,
                  int Cp, int accumulate, hipStream_t st) {
  const int Cv = Cp / 4;
  const int padT = (H - 2 * h) / 2, padL = (W - 2 * w) / 2;
  MIMO_ST_DISPATCH(dt, T, hipLaunchKernelGGL(up_bwd_kernel<T>, pq_grid(Cv, (int64_t)N * h * w, kBlocksUpBwd), dim3(256), 0, st,
                                             (const T*)dxpad, ldp, choff, (T*)da, ldda, N, H, W, h, w, padT, padL, Cv, accumulate));
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// ---------------------------------------------------------------------------------------
// BatchNorm + ReLU backward
// ---------------------------------------------------------------------------------------
// value and gradient of the element-wise NLL.  The clamp acts on the value only
// (losses.py:153-155 clamps in place under no_grad), so d/dlog keeps the unclamped exp.
__device__ __forceinline__ float nll_value(int kind, float d, float lp, float eps_min, float eps_max) {
  const float sc = fminf(fmaxf(expf(lp), eps_min), eps_max);
  return kind == MIMO_LOSS_LAPLACE_NLL ? logf(sc) + fabsf(d) / sc : logf(sc) + d * d / sc;
}
__device__ __forceinline__ void nll_grad(int kind, float d, float lp, float eps_min, float eps_max, float* gmu, float* glp) {
  const float e = expf(lp);
  const float sc = fminf(fmaxf(e, eps_min), eps_max);
  if (kind == MIMO_LOSS_LAPLACE_NLL) {
    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    *gmu = sgn / sc;
    *glp = (1.f / sc - fabsf(d) / (sc * sc)) * e;
  } else {
    *gmu = 2.f * d / sc;
    *glp = (1.f / sc - d * d / (sc * sc)) * e;
  }
}

// dy = (gradient arriving at the activation) * dropout mask * [relu input > 0], evaluated on the fly by
// both passes (never stored); the arriving gradient per GradSrc (elementwise.h)
struct HeadLane {  // GS_HEAD per-thread constants: this channel quad's 1x1 weights, dloss[s] / count
  float4 w0, w1;
  float coef;
};
__device__ __forceinline__ HeadLane head_lane(const HeadGrad& h, int q, bool active) {
  HeadLane l;
  l.w0 = l.w1 = f4zero();
  l.coef = h.dloss ? h.dloss[h.s] * h.inv_count : 0.f;
  if (active) {
    const int c0 = 4 * q;
    float* a0 = reinterpret_cast<float*>(&l.w0);
    float* a1 = reinterpret_cast<float*>(&l.w1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a0[j] = c0 + j < h.C ? h.w[c0 + j] : 0.f;
      a1[j] = c0 + j < h.C ? h.w[h.C + c0 + j] : 0.f;
    }
  }
  return l;
}
// dlogit of pixel p (Co == 2: mean, log-scale), head_bwd_kernel's arithmetic
__device__ __forceinline__ void head_dlogit(const HeadGrad& h, const HeadLane& l, int p, float* d0, float* d1) {
  const int n = p / h.HW;
  const int yx = p - n * h.HW;
  const int64_t obase = (((int64_t)n * h.S + h.s) * 2) * h.HW + yx;
  float a = h.dout ? h.dout[obase] : 0.f, b = h.dout ? h.dout[obase + h.HW] : 0.f;
  if (h.dloss) {
    const int64_t src = h.perm ? h.perm[(int64_t)h.s * h.N + n] : n;
    const float mk = h.mask ? h.mask[src * h.HW + yx] : 1.f;
    const float mu = h.out[obase], lp = h.out[obase + h.HW], y = h.label[src * h.HW + yx];
    float gm, gl;
    nll_grad(h.kind, mu - y, lp, h.eps_min, h.eps_max, &gm, &gl);
    a += l.coef * mk * gm;
    b += l.coef * mk * gl;
  }
  *d0 = a;
  *d1 = b;
}

__device__ __forceinline__ float4 relu_gate4(float4 g, float4 v, float4 sc, float4 sh) {
  g.x = fmaf(v.x, sc.x, sh.x) > 0.f ? g.x : 0.f;
  g.y = fmaf(v.y, sc.y, sh.y) > 0.f ? g.y : 0.f;
  g.z = fmaf(v.z, sc.z, sh.z) > 0.f ? g.z : 0.f;
  g.w = fmaf(v.w, sc.w, sh.w) > 0.f ? g.w : 0.f;
  return g;
}

// GS_POOL: one thread owns the 2x2 cell (cy, cx) of image n — as pool_bwd_kernel did: one folded read of the pooled
// gradient, the four pre-activation values once, the window's activations re-formed with bn_relu_pool_fwd_kernel's
// arithmetic, the first maximum in scan order takes the gradient, + the folded skip gradient of each pixel.  Cells of the
// last row / column of an odd-sized image hold one or two pixels and no window.  Returns the valid pixels as a bit mask,
// their pre-activation values in zz[] and their dy (dropout multiplier and ReLU gate applied) in g[].
template <typename TZ, typename TA>
__device__ __forceinline__ int pool_cell4(const GradSrc& s, const TZ* z, int ldz, int n, int cy, int cx, int H, int W, int q,
                                          float4 sc, float4 sh, bool masked, float4 m, float4 zz[4], float4 g[4]) {
  const int Hp = H / 2, Wp = W / 2, y0 = 2 * cy, x0 = 2 * cx;
  const bool win = cy < Hp && cx < Wp;
  const int valid = win ? 15 : ((1 | (x0 + 1 < W ? 2 : 0)) | (y0 + 1 < H ? (4 | (x0 + 1 < W ? 8 : 0)) : 0));
  const TZ* zs = z + (((size_t)n * H + y0) * W + x0) * ldz + 4 * q;
#pragma unroll
  for (int j = 0; j < 4; ++j) zz[j] = (valid >> j) & 1 ? ld4(zs + ((size_t)(j >> 1) * W + (j & 1)) * ldz) : f4zero();
  float4 gp = f4zero();
  if (win) gp = fold_read((const TA*)s.dxpad, s.ldp, n, cy, cx, Hp, Wp, s.choff + 4 * q);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    g[j] = (s.skip && ((valid >> j) & 1)) ? fold_read((const TA*)s.skip, s.ldsk, n, y0 + (j >> 1), x0 + (j & 1), H, W, s.skoff + 4 * q)
                                           : f4zero();
  if (win) {
    auto act = [&](float v, float k, float b, float mm) { return fmaxf(fmaf(v, k, b), 0.f) * mm; };
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // route + skip: pool_bwd_kernel's operands in its order
      g[j].x = route_max(gp.x, act(zz[0].x, sc.x, sh.x, m.x), act(zz[1].x, sc.x, sh.x, m.x), act(zz[2].x, sc.x, sh.x, m.x), act(zz[3].x, sc.x, sh.x, m.x), j) + g[j].x;
      g[j].y = route_max(gp.y, act(zz[0].y, sc.y, sh.y, m.y), act(zz[1].y, sc.y, sh.y, m.y), act(zz[2].y, sc.y, sh.y, m.y), act(zz[3].y, sc.y, sh.y, m.y), j) + g[j].y;
      g[j].z = route_max(gp.z, act(zz[0].z, sc.z, sh.z, m.z), act(zz[1].z, sc.z, sh.z, m.z), act(zz[2].z, sc.z, sh.z, m.z), act(zz[3].z, sc.z, sh.z, m.z), j) + g[j].z;
      g[j].w = route_max(gp.w, act(zz[0].w, sc.w, sh.w, m.w), act(zz[1].w, sc.w, sh.w, m.w), act(zz[2].w, sc.w, sh.w, m.w), act(zz[3].w, sc.w, sh.w, m.w), j) + g[j].w;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (masked) {
      g[j].x *= m.x;
      g[j].y *= m.y;
      g[j].z *= m.z;
      g[j].w *= m.w;
    }
    g[j] = relu_gate4(g[j], zz[j], sc, sh);
  }
  return valid;
}

template <int SRC, typename TZ, typename TA>
__device__ __forceinline__ float4 relu_grad4(const GradSrc& s, const HeadLane& hl, const float* mask, int C, int p,
                                             const PixIter& it, int q, int H, int W, float4 v, float4 sc, float4 sh, float* d0,
                                             float* d1) {
  static_assert(SRC != GS_POOL, "pooled tensors: pool_cell4");
  float4 g;
  const int n = it.n;
  float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
  if (mask) m = mask4(mask, n, C, 4 * q);
  if constexpr (SRC == GS_PLAIN) {
    g = ld4((const TA*)s.da + (size_t)p * s.ldda + 4 * q);
  } else if constexpr (SRC == GS_FOLD) {
    g = fold_read((const TA*)s.dxpad, s.ldp, n, it.y, it.x, H, W, 4 * q);
  } else {
    head_dlogit(s.head, hl, p, d0, d1);
    g = f4zero();
    g.x = fmaf(*d1, hl.w1.x, fmaf(*d0, hl.w0.x, g.x));
    g.y = fmaf(*d1, hl.w1.y, fmaf(*d0, hl.w0.y, g.y));
    g.z = fmaf(*d1, hl.w1.z, fmaf(*d0, hl.w0.z, g.z));
    g.w = fmaf(*d1, hl.w1.w, fmaf(*d0, hl.w0.w, g.w));
  }
  if (mask) {
    g.x *= m.x;
    g.y *= m.y;
    g.z *= m.z;
    g.w *= m.w;
  }
  return relu_gate4(g, v, sc, sh);
}

template <typename TZ, typename TA, int SRC>
__global__ __launch_bounds__(kBnReduceThreads) void bnrelu_bwd_reduce_kernel(
    const GradSrc src, const TZ* __restrict__ z, int ldz,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ mask, int C, int Cv, int N, int H, int W,
    float* __restrict__ partial) {
  __shared__ float4 red[kBnReduceThreads];
  const PQ t = pixquad<kBnReduceThreads>(Cv);
  const int Cp = 4 * Cv;
  float4 a1 = f4zero(), a2 = f4zero();
  float4 hw0 = f4zero(), hw1 = f4zero();  // GS_HEAD: the head's weight gradient (this quad's channels x 2 logits), bias gradient
  float hb0 = 0.f, hb1 = 0.f;
  HeadLane hl = {};
  if constexpr (SRC == GS_HEAD) hl = head_lane(src.head, t.q, t.active);
  if (t.active) {
    const float4 sc = ld4(scale + 4 * t.q), sh = ld4(shift + 4 * t.q), mu = ld4(mean + 4 * t.q), is = ld4(invstd + 4 * t.q);
    if constexpr (SRC == GS_POOL) {
      const int Hc = (H + 1) / 2, Wc = (W + 1) / 2, P = N * Hc * Wc;
      PixIter it = pix_iter(t.p, t.pstep, Hc, Wc);
      for (int p = t.p; p < P; p += t.pstep, pix_next(it, Hc, Wc)) {
        float4 zz[4], g[4];
        const float4 m = mask ? mask4(mask, it.n, C, 4 * t.q) : make_float4(1.f, 1.f, 1.f, 1.f);
        const int valid = pool_cell4<TZ, TA>(src, z, ldz, it.n, it.y, it.x, H, W, t.q, sc, sh, mask != nullptr, m, zz, g);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if ((valid >> j) & 1) {
            a1 = f4add(a1, g[j]);
            a2.x += g[j].x * (zz[j].x - mu.x) * is.x;
            a2.y += g[j].y * (zz[j].y - mu.y) * is.y;
            a2.z += g[j].z * (zz[j].z - mu.z) * is.z;
            a2.w += g[j].w * (zz[j].w - mu.w) * is.w;
          }
      }
    } else {
      const int P = N * H * W;
      PixIter it = pix_iter(t.p, t.pstep, H, W);
      for (int p = t.p; p < P; p += t.pstep, pix_next(it, H, W)) {
        const float4 v = ld4(z + (size_t)p * ldz + 4 * t.q);
        float d0 = 0.f, d1 = 0.f;
        const float4 g = relu_grad4<SRC, TZ, TA>(src, hl, mask, C, p, it, t.q, H, W, v, sc, sh, &d0, &d1);
        a1 = f4add(a1, g);
        a2.x += g.x * (v.x - mu.x) * is.x;
        a2.y += g.y * (v.y - mu.y) * is.y;
        a2.z += g.z * (v.z - mu.z) * is.z;
        a2.w += g.w * (v.w - mu.w) * is.w;
        if constexpr (SRC == GS_HEAD) {  // the head's input is this tensor's activation (no dropout on this path)
          const float4 av = bn_relu4(v, sc, sh);
          hw0.x = fmaf(d0, av.x, hw0.x);
          hw0.y = fmaf(d0, av.y, hw0.y);
          hw0.z = fmaf(d0, av.z, hw0.z);
          hw0.w = fmaf(d0, av.w, hw0.w);
          hw1.x = fmaf(d1, av.x, hw1.x);
          hw1.y = fmaf(d1, av.y, hw1.y);
          hw1.z = fmaf(d1, av.z, hw1.z);
          hw1.w = fmaf(d1, av.w, hw1.w);
          hb0 += d0;
          hb1 += d1;
        }
      }
    }
  }
  const float4 s1 = quad_block_sum(a1, t, red);
  const float4 s2 = quad_block_sum(a2, t, red);
  if (t.pl == 0 && t.q < Cv) {
    float* row = partial + (size_t)blockIdx.x * 2 * Cp;
    st4(row + 4 * t.q, s1);
    st4(row + Cp + 4 * t.q, s2);
  }
  if constexpr (SRC == GS_HEAD) {  // partial row as head_bwd_kernel writes it: [2][Cp] weight gradient, [2] bias gradient
    float* row = src.head.partial + (size_t)blockIdx.x * (2 * Cp + 2);
    const float4 w0 = quad_block_sum(hw0, t, red);
    const float4 w1 = quad_block_sum(hw1, t, red);
    const float4 sb = quad_block_sum(make_float4(hb0, hb1, 0.f, 0.f), t, red);
    if (t.pl == 0 && t.q < Cv) {
      st4(row + 4 * t.q, w0);
      st4(row + Cp + 4 * t.q, w1);
    }
    if (t.pl == 0 && t.q == 0) {
      row[2 * Cp] = sb.x;
      row[2 * Cp + 1] = sb.y;
    }
  }
}

// the fused sources exist for fp32 storage (TZ = TA = float) only
static int check_grad_src(const GradSrc& src, int dta, int dtz) {
  if (src.kind < GS_PLAIN || src.kind > GS_HEAD || (src.kind >= GS_POOL && (dta != ST_F32 || dtz != ST_F32)) ||
      (src.kind == GS_HEAD && src.head.Co != 2)) {
    set_error("BatchNorm backward: gradient source %d unsupported here", src.kind);
    return MIMO_ERR_INVALID;
  }
  return MIMO_OK;
}

int bnrelu_bwd_reduce_launch(const GradSrc& src, int dta, const void* z, int dtz, int ldz,
                             const float* scale, const float* shift, const float* mean, const float* invstd,
                             const float* mask, int C, int Cp, int N, int H, int W, float* partial, int* rows,
                             hipStream_t st) {
  MIMO_TRY(check_grad_src(src, dta, dtz));
  const int Cv = Cp / 4;
  const int64_t units = src.kind == GS_POOL ? (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) : (int64_t)N * H * W;  // cells / pixels
  // (the fused sources hold more registers: 4 / 5 waves per SIMD = one 1024-thread workgroup per CU, a single round)
  const dim3 grid = pq_grid(Cv, units, src.kind >= GS_POOL ? 256 : kBlocksBnReduce, kBnReduceThreads);
  *rows = grid.x;
#define REDUCE_LAUNCH(TZ, TA, SRC)                                                                                              \
  hipLaunchKernelGGL((bnrelu_bwd_reduce_kernel<TZ, TA, SRC>), grid, dim3(kBnReduceThreads), 0, st, src, (const TZ*)z, ldz, scale, \
                     shift, mean, invstd, mask, C, Cv, N, H, W, partial)
  if (src.kind == GS_POOL) {
    REDUCE_LAUNCH(float, float, GS_POOL);
  } else if (src.kind == GS_HEAD) {
    REDUCE_LAUNCH(float, float, GS_HEAD);
  } else if (src.kind == GS_PLAIN) {
    MIMO_ST_DISPATCH2(dtz, dta, TZ, TA, REDUCE_LAUNCH(TZ, TA, GS_PLAIN))
  } else {
    MIMO_ST_DISPATCH2(dtz, dta, TZ, TA, REDUCE_LAUNCH(TZ, TA, GS_FOLD))
  }
#undef REDUCE_LAUNCH
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

__global__ void bn_bwd_finalize_kernel(const double* __restrict__ sums, int chunks, int C, int Cp, double count,
                                       int training, float* __restrict__ c1, float* __restrict__ c2,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ double red[512];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  double s1, s2;
  chunk_total2(sums + c, sums + Cp + c, (size_t)2 * Cp, chunks, c < Cp, red, &s1, &s2);
  if (c >= Cp || threadIdx.x >= 64) return;
  c1[c] = training ? (float)(s1 / count) : 0.f;
  c2[c] = training ? (float)(s2 / count) : 0.f;
  if (c < C) {
    if (dgamma) dgamma[c] = (float)s2;
    if (dbeta) dbeta[c] = (float)s1;
  }
}

// rowsum + bn_bwd_finalize in one launch: partial rows [rows][2][Cp] from bnrelu_bwd_reduce
__global__ __launch_bounds__(kColsumThreads) void bn_bwd_stats_kernel(const float* __restrict__ partial, int rows, int C,
                                                                      int Cp, double count, int training,
                                                                      float* __restrict__ c1, float* __restrict__ c2,
                                                                      float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                      float* __restrict__ dbias_zero,
                                                                      double* __restrict__ scratch, int scratch_cols,
                                                                      int* __restrict__ tickets, int* __restrict__ status) {
  __shared__ double red[2048];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  double s1, s2;
  if (!grid_colsum2(partial, rows, (size_t)2 * Cp, c, Cp + c, c < Cp, true, red, scratch, scratch_cols, tickets, &s1, &s2)) return;
  if (c >= Cp || threadIdx.x >= 64) return;
  if (status && c < C && !(isfinite(s1) && isfinite(s2))) atomicOr(status, kStatusBwdStats);
  c1[c] = training ? (float)(s1 / count) : 0.f;
  c2[c] = training ? (float)(s2 / count) : 0.f;
  if (c < C) {
    if (dgamma) dgamma[c] = (float)s2;
    if (dbeta) dbeta[c] = (float)s1;
    // training-mode BatchNorm removes any per-channel shift of its input: the gradient of the convolution bias
    // in front of it is exactly zero (the reference's autograd produces rounding noise around zero)
    if (dbias_zero) dbias_zero[c] = 0.f;
  }
}

int bn_bwd_stats_launch(const float* partial, int rows, int C, int Cp, int64_t count, int training, float* c1, float* c2,
                        float* dgamma, float* dbeta, float* dbias_zero, const ColsumScratch& cs, hipStream_t st) {
  const int groups = ceil_div(Cp, 64);
  hipLaunchKernelGGL(bn_bwd_stats_kernel, dim3(groups, colsum_chunks(rows)), dim3(kColsumThreads), 0, st, partial, rows, C, Cp,
                     (double)count, training, c1, c2, dgamma, dbeta, dbias_zero, cs.sums, groups * 64, cs.tickets, cs.status);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

int bn_bwd_finalize_launch(const double* sums, int chunks, int C, int Cp, int64_t count, int training, float* c1,
                           float* c2, float* dgamma, float* dbeta, hipStream_t st) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(ceil_div(Cp, 64)), dim3(256), 0, st, sums, chunks, C, Cp, (double)count,
                     training, c1, c2, dgamma, dbeta);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// Split storage of a gradient tensor for the bf16-pair MFMA kernels (data and weight gradient): per
// pixel, per 32-channel chunk, [hi: 32 bf16 | lo: 32 bf16] with v ~= hi + lo (a last partial chunk of r
// channels is [hi r | lo r]) — the same 4*Cp bytes as fp32, and one contiguous 128-byte line per
// (pixel, chunk), which is exactly the LDS row image of the data-gradient kernel.  The split is the one
// the convolution loaders used to do on every read (hi = RNE(v), lo = RNE(v - hi)).
typedef __bf16 bf16x4_ew __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_split4(float* base, size_t p, int Cp, int q, float4 r) {
  bf16x4_ew hi, lo;
  hi[0] = (__bf16)r.x;
  hi[1] = (__bf16)r.y;
  hi[2] = (__bf16)r.z;
  hi[3] = (__bf16)r.w;
  lo[0] = (__bf16)(r.x - (float)hi[0]);
  lo[1] = (__bf16)(r.y - (float)hi[1]);
  lo[2] = (__bf16)(r.z - (float)hi[2]);
  lo[3] = (__bf16)(r.w - (float)hi[3]);
  const int ch = 4 * q, chunk = ch >> 5, rc = min(32, Cp - 32 * chunk);
  unsigned char* d = reinterpret_cast<unsigned char*>(base) + p * (size_t)Cp * 4 + chunk * 128 + (ch & 31) * 2;
  *reinterpret_cast<bf16x4_ew*>(d) = hi;
  *reinterpret_cast<bf16x4_ew*>(d + 2 * rc) = lo;
}

// max over the workgroup of a non-negative per-thread value -> the workgroup's own slot of `slots` (a plain store, no
// atomic; one barrier at the end of the kernel).  Whoever needs the tensor's maximum reduces the slots (wg_dz_absmax,
// common.h).  Earlier versions kept ONE word per tensor: an atomic max per workgroup (+4 us per launch: ~1800 same-address
// atomics queue), then per wave behind a relaxed load of the word (+25 us: the loads queue too — 4.66 -> 5.25 ms per step
// at 4 images per GPU); per-wave slots cost the consumers 4 x the loads (profiles/r05/wgrad_two_mfma.txt item 5).
__device__ __forceinline__ void block_absmax_store(float* slots, float v) {
  __shared__ float wm[16];
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) v = fmaxf(v, __shfl_xor(v, d));
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = wm[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, wm[i]);
    slots[blockIdx.y * gridDim.x + blockIdx.x] = m;
  }
}
__device__ __forceinline__ float absmax4(float m, float4 r) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
}

__global__ void split_pairs_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cv, int64_t P,
                                   float* __restrict__ absmax) {
  const PQ t = pixquad(Cv);
  float amax = 0.f;
  if (t.active)
    for (int64_t p = t.p; p < P; p += t.pstep) {
      const float4 r = ld4(src + (size_t)p * 4 * Cv + 4 * t.q);
      st_split4(dst, (size_t)p, 4 * Cv, t.q, r);
      amax = absmax4(amax, r);
    }
  if (absmax) block_absmax_store(absmax, amax);
}

int split_pairs_launch(const float* src, float* dst, int64_t P, int Cp, hipStream_t st, float* absmax, int* absmax_n) {
  const dim3 grid = pq_grid(Cp / 4, P, 2048);
  if (absmax && (int)(grid.x * grid.y) > kDzMaxSlots) {
    set_error("split_pairs: %d x %d workgroups exceed the max |dz| slots", (int)grid.x, (int)grid.y);
    return MIMO_ERR_INVALID;
  }
  if (absmax_n) *absmax_n = (int)(grid.x * grid.y);
  hipLaunchKernelGGL(split_pairs_kernel, grid, dim3(256), 0, st, src, dst, Cp / 4, P, absmax);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// (plain / folded sources: held to 72 registers = 7 waves per SIMD, the occupancy round 3's grid scan found best — the
// max |dz| tracking of round 5 had pushed the folded instance to 74 and cost the class 7 %)
#ifndef MIMO_APPLY_MINWAVES
#define MIMO_APPLY_MINWAVES 7
#endif
#ifndef MIMO_APPLY_REVERSE
#define MIMO_APPLY_REVERSE 1  // 0: front to back (A/B builds)
#endif
#ifndef MIMO_APPLY_ABSMAX
#define MIMO_APPLY_ABSMAX 1  // 0: A/B build without the max |dz| tracking (the two-MFMA weight gradient then reads garbage)
#endif
template <typename TZ, typename TA, int SRC>
__global__ __launch_bounds__(256, (SRC == GS_PLAIN || SRC == GS_FOLD) ? MIMO_APPLY_MINWAVES : 1) void bn_bwd_apply_kernel(const GradSrc src, const TZ* __restrict__ z, int ldz, const float* __restrict__ scale,
                                    const float* __restrict__ shift, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ mask, int C,
                                    const float* __restrict__ c1, const float* __restrict__ c2, int Cv, int N, int H,
                                    int W, TA* __restrict__ dz, int split_out, float* __restrict__ partial,
                                    float* __restrict__ absmax) {
  __shared__ float4 red[256];
  const PQ t = pixquad(Cv);
  const int Cp = 4 * Cv;
  float4 acc = f4zero();
  float amax = 0.f;  // max |dz| of this thread's elements (absmax != nullptr: the two-MFMA weight gradient scales dz by it)
  HeadLane hl = {};
  if constexpr (SRC == GS_HEAD) hl = head_lane(src.head, t.q, t.active);
  if (t.active) {
    const float4 sc = ld4(scale + 4 * t.q), sh = ld4(shift + 4 * t.q), mu = ld4(mean + 4 * t.q), is = ld4(invstd + 4 * t.q);
    const float4 k1 = ld4(c1 + 4 * t.q), k2 = ld4(c2 + 4 * t.q);
    auto emit = [&](int p, float4 g, float4 v) {
      float4 r;
      r.x = sc.x * (g.x - k1.x - (v.x - mu.x) * is.x * k2.x);
      r.y = sc.y * (g.y - k1.y - (v.y - mu.y) * is.y * k2.y);
      r.z = sc.z * (g.z - k1.z - (v.z - mu.z) * is.z * k2.z);
      r.w = sc.w * (g.w - k1.w - (v.w - mu.w) * is.w * k2.w);
      if (split_out)  // fp32 storage only: bf16 (hi, lo) pair records
        st_split4(reinterpret_cast<float*>(dz), p, Cp, t.q, r);
      else
        st4(dz + (size_t)p * Cp + 4 * t.q, r);
      acc = f4add(acc, r);
      if (MIMO_APPLY_ABSMAX) amax = absmax4(amax, r);
    };
    if constexpr (SRC == GS_POOL) {
      const int Hc = (H + 1) / 2, Wc = (W + 1) / 2, P = N * Hc * Wc;
      PixIter it = pix_iter(t.p, t.pstep, Hc, Wc);
      for (int p = t.p; p < P; p += t.pstep, pix_next(it, Hc, Wc)) {
        float4 zz[4], g[4];
        const float4 m = mask ? mask4(mask, it.n, C, 4 * t.q) : make_float4(1.f, 1.f, 1.f, 1.f);
        const int valid = pool_cell4<TZ, TA>(src, z, ldz, it.n, it.y, it.x, H, W, t.q, sc, sh, mask != nullptr, m, zz, g);
        const int p0 = (it.n * H + 2 * it.y) * W + 2 * it.x;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if ((valid >> j) & 1) emit(p0 + (j >> 1) * W + (j & 1), g[j], zz[j]);
      }
    } else {
      const int P = N * H * W;
#if MIMO_APPLY_REVERSE
      // back to front: the statistics pass in front of this one read the same two tensors front to back, so their LAST
      // part is what the last-level cache still holds
      if (t.p < P) {
        const int p_last = t.p + (P - 1 - t.p) / t.pstep * t.pstep;
        PixIter it = pix_iter(p_last, t.pstep, H, W);
        for (int p = p_last; p >= 0; p -= t.pstep, pix_prev(it, H, W)) {
          const float4 v = ld4(z + (size_t)p * ldz + 4 * t.q);
          float d0, d1;
          emit(p, relu_grad4<SRC, TZ, TA>(src, hl, mask, C, p, it, t.q, H, W, v, sc, sh, &d0, &d1), v);
        }
      }
#else
      PixIter it = pix_iter(t.p, t.pstep, H, W);
      for (int p = t.p; p < P; p += t.pstep, pix_next(it, H, W)) {
        const float4 v = ld4(z + (size_t)p * ldz + 4 * t.q);
        float d0, d1;
        emit(p, relu_grad4<SRC, TZ, TA>(src, hl, mask, C, p, it, t.q, H, W, v, sc, sh, &d0, &d1), v);
      }
#endif
    }
  }
  if (MIMO_APPLY_ABSMAX && absmax) block_absmax_store(absmax, amax);
  if (!partial) return;  // training mode: the bias gradient is exactly zero, no column sums wanted
  const float4 s = quad_block_sum(acc, t, red);
  if (t.pl == 0 && t.q < Cv) st4(partial + (size_t)blockIdx.x * Cp + 4 * t.q, s);
}

int bn_bwd_apply_launch(const GradSrc& src, int dta, const void* z, int dtz, int ldz,
                        const float* scale, const float* shift, const float* mean, const float* invstd, const float* mask,
                        int C, const float* c1, const float* c2, int Cp, int N, int H, int W, void* dz, int split_out,
                        float* partial, int* rows, hipStream_t st, float* absmax, int* absmax_n, hipEvent_t done) {
  MIMO_TRY(check_grad_src(src, dta, dtz));
  const int Cv = Cp / 4;
  const int64_t units = src.kind == GS_POOL ? (int64_t)N * ((H + 1) / 2) * ((W + 1) / 2) : (int64_t)N * H * W;
  const dim3 grid = pq_grid(Cv, units, src.kind == GS_POOL ? 1024 : src.kind == GS_HEAD ? 1280 : kBlocksBnBwd);  // 4 / 5 / 7 per CU
  *rows = grid.x;
  if (absmax_n) *absmax_n = (int)(grid.x * grid.y);  // one slot per workgroup
  if (absmax && (int)(grid.x * grid.y) > kDzMaxSlots) {
    set_error("bn_bwd_apply: %d x %d workgroups exceed the max |dz| slots", (int)grid.x, (int)grid.y);
    return MIMO_ERR_INVALID;
  }
  if (split_out && dta != ST_F32) {
    set_error("bn_bwd_apply: pair-split dz exists for fp32 storage only");
    return MIMO_ERR_INVALID;
  }
  // `done`: recorded when THIS kernel completes, carried by the launch itself (hipExtLaunchKernelGGL stop event) — a separate
  // hipEventRecord behind the kernel costs the stream 2.9-5 us per hand-off (scripts/micro/event_cost.hip), this one ~1
#define APPLY_LAUNCH(TZ, TA, SRC)                                                                                              \
  hipExtLaunchKernelGGL((bn_bwd_apply_kernel<TZ, TA, SRC>), grid, dim3(256), 0, st, nullptr, done, 0, src, (const TZ*)z, ldz, scale, \
                        shift, mean, invstd, mask, C, c1, c2, Cv, N, H, W, (TA*)dz, split_out, partial, absmax)
  if (src.kind == GS_POOL) {
    APPLY_LAUNCH(float, float, GS_POOL);
  } else if (src.kind == GS_HEAD) {
    APPLY_LAUNCH(float, float, GS_HEAD);
  } else if (src.kind == GS_PLAIN) {
    MIMO_ST_DISPATCH2(dtz, dta, TZ, TA, APPLY_LAUNCH(TZ, TA, GS_PLAIN))
  } else {
    MIMO_ST_DISPATCH2(dtz, dta, TZ, TA, APPLY_LAUNCH(TZ, TA, GS_FOLD))
  }
#undef APPLY_LAUNCH
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// ---------------------------------------------------------------------------------------
// 1x1 head and the NLL losses
// ---------------------------------------------------------------------------------------
// G lanes share a pixel (G = power of two >= channel quads): every lane loads one float4 of the pixel's
// channel row — a wave reads 64 consecutive float4, fully coalesced — multiplies it with its slice of the
// 1x1 weights, and a butterfly of G-lane shuffles completes the Co dot products.  (One thread per pixel
// reading its own 128-byte row touched 64 cache lines per load instruction and ran at 1 TB/s.)
template <int G, int CO, typename T>
__global__ void head_fwd_kernel(const T* __restrict__ a, int lda, const float* __restrict__ w,
                                const float* __restrict__ bias, int C, int Cp, int Co, int N, int S, int s, int HW,
                                float* __restrict__ out, int* __restrict__ status, const float* __restrict__ in_scale,
                                const float* __restrict__ in_shift) {
  // in_scale != nullptr: `a` is the pre-activation tensor of the decoder's last convolution, its BatchNorm + ReLU is applied here
  // CO = compile-time bound of Co (2: one target, 4: the evidential head, 8: the rest); UN pixels per thread and
  // iteration, their loads issued together (one 16-byte load in flight per thread ran at 3.4 TB/s)
  const int g = threadIdx.x % G, pl = threadIdx.x / G;
  constexpr int PPB = 256 / G, UN = 4;
  float wq[CO][4], bs[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    bs[co] = co < Co ? bias[co] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) wq[co][j] = (co < Co && 4 * g + j < C) ? w[co * C + 4 * g + j] : 0.f;
  }
  const bool has = 4 * g < Cp;
  const bool fin = in_scale != nullptr;
  const float4 isc = (fin && has) ? ld4(in_scale + 4 * g) : f4zero(), ish = (fin && has) ? ld4(in_shift + 4 * g) : f4zero();
  const int64_t P = (int64_t)N * HW, stride = (int64_t)gridDim.x * PPB;
  for (int64_t p0 = (int64_t)blockIdx.x * PPB + pl; p0 < P; p0 += stride * UN) {
    float4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t p = p0 + u * stride;
      v[u] = (has && p < P) ? ld4(a + p * lda + 4 * g) : f4zero();
      if (fin) v[u] = bn_relu4(v[u], isc, ish);  // (lanes without channels: relu(0 * 0 + 0) = 0)
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t p = p0 + u * stride;
      float acc[CO];
#pragma unroll
      for (int co = 0; co < CO; ++co)
        acc[co] = fmaf(v[u].x, wq[co][0], fmaf(v[u].y, wq[co][1], fmaf(v[u].z, wq[co][2], v[u].w * wq[co][3])));
#pragma unroll
      for (int off = 1; off < G; off <<= 1)
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] += __shfl_xor(acc[co], off);
      if (p < P) {
        const int n = (int)(p / HW);
        const int yx = (int)(p - (int64_t)n * HW);
#pragma unroll
        for (int co = 0; co < CO; ++co)
          if (co < Co && co % G == g) {
            const float o = acc[co] + bs[co];
            out[(((int64_t)n * S + s) * Co + co) * HW + yx] = o;
            if (status && !isfinite(o)) atomicOr(status, kStatusLogits);
          }
      }
    }
  }
}

int head_fwd_launch(const void* a, int dt, int lda, const float* w, const float* bias, int C, int Co, int N, int S, int s,
                    int HW, float* out, hipStream_t st, int* status, const float* in_scale, const float* in_shift) {
  if (Co > kMaxHeadOut || C > 256) {
    set_error("head: out_channels %d > %d or filter_base_count %d > 256 unsupported", Co, kMaxHeadOut, C);
    return MIMO_ERR_INVALID;
  }
  const int Cp = pad_channels(C), Cv = Cp / 4;
  int G = 2;
  while (G < Cv) G <<= 1;
  const int64_t P = (int64_t)N * HW;
  const int blocks = (int)std::min<int64_t>(ceil_div64(P, (256 / G) * 4), 4096);
#define HEAD_LAUNCH2(GG, CC)                                                                                         \
  MIMO_ST_DISPATCH(dt, T, hipLaunchKernelGGL((head_fwd_kernel<GG, CC, T>), dim3(blocks), dim3(256), 0, st, (const T*)a, lda, w, bias, \
                                             C, Cp, Co, N, S, s, HW, out, status, in_scale, in_shift))
#define HEAD_LAUNCH(GG)             \
  if (Co <= 2) {                    \
    HEAD_LAUNCH2(GG, 2);            \
  } else if (Co <= 4) {             \
    HEAD_LAUNCH2(GG, 4);            \
  } else {                          \
    HEAD_LAUNCH2(GG, kMaxHeadOut);  \
  }
  switch (G) {
    case 2: HEAD_LAUNCH(2); break;
    case 4: HEAD_LAUNCH(4); break;
    case 8: HEAD_LAUNCH(8); break;
    case 16: HEAD_LAUNCH(16); break;
    case 32: HEAD_LAUNCH(32); break;
    default: HEAD_LAUNCH(64); break;
  }
#undef HEAD_LAUNCH
#undef HEAD_LAUNCH2
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

__global__ void loss_fwd_kernel(const float* __restrict__ out, const float* __restrict__ label,
                                const float* __restrict__ mask, const int64_t* __restrict__ perm, int N, int S, int Co,
                                int HW, int kind, float eps_min, float eps_max, float* __restrict__ partial) {
  __shared__ float red[256];
  const int s = blockIdx.y;
  const int Ct = Co / 2;
  const int64_t total = (int64_t)N * Ct * HW;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int yx = (int)(i % HW);
    const int64_t r = i / HW;
    const int tch = (int)(r % Ct);
    const int n = (int)(r / Ct);
    const int64_t src = perm ? perm[(int64_t)s * N + n] : n;
    const float* o = out + (((int64_t)n * S + s) * Co) * HW + yx;
    const float mu = o[(int64_t)tch * HW], lp = o[(int64_t)(Ct + tch) * HW];
    const float y = label[(src * Ct + tch) * HW + yx];
    float v = nll_value(kind, mu - y, lp, eps_min, eps_max);
    if (mask) v *= mask[src * HW + yx];
    acc += v;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(size_t)s * gridDim.x + blockIdx.x] = red[0];
}

int loss_fwd_launch(const float* out, const float* label, const float* mask, const int64_t* perm, int N, int S, int Co,
                    int HW, int kind, float eps_min, float eps_max, float* partial, int* blocks, hipStream_t st) {
  const int64_t total = (int64_t)N * (Co / 2) * HW;
  const int nb = (int)std::min<int64_t>(ceil_div64(total, 256), 512);
  *blocks = nb;
  hipLaunchKernelGGL(loss_fwd_kernel, dim3(nb, S), dim3(256), 0, st, out, label, mask, perm, N, S, Co, HW, kind, eps_min,
                     eps_max, partial);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

__global__ void loss_finalize_kernel(const float* __restrict__ partial, int S, int blocks, double count,
                                     float* __restrict__ loss_out) {
  __shared__ double red[256];
  const int s = blockIdx.x;
  double acc = 0.0;
  for (int i = threadIdx.x; i < blocks; i += blockDim.x) acc += (double)partial[(size_t)s * blocks + i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[s] = (float)(red[0] / count);
}

int loss_finalize_launch(const float* partial, int S, int blocks, double count, float* loss_out, hipStream_t st) {
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(S), dim3(256), 0, st, partial, S, blocks, count, loss_out);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// CO = compile-time bound of Co (as in head_fwd_kernel); UN pixels per thread and iteration, loads issued together (the
// grid is capped at kEwMaxBlocks partial rows = 16 waves per CU: one pixel at a time left 16 KB in flight per CU)
template <int CO, typename T>
__global__ void head_bwd_kernel(const T* __restrict__ a, int lda, const float* __restrict__ w, int C, int Cv, int Co,
                                int N, int S, int s, int HW, const float* __restrict__ out,
                                const float* __restrict__ dout, const float* __restrict__ dloss,
                                const float* __restrict__ label, const float* __restrict__ mask,
                                const int64_t* __restrict__ perm, int kind, float eps_min, float eps_max, float inv_count,
                                T* __restrict__ da, float* __restrict__ partial, const float* __restrict__ in_scale,
                                const float* __restrict__ in_shift) {
  __shared__ float4 red[256];
  constexpr int UN = 4;
  const PQ t = pixquad(Cv);
  const bool fin = in_scale != nullptr;
  const float4 isc = (fin && t.active) ? ld4(in_scale + 4 * t.q) : f4zero(), ish = (fin && t.active) ? ld4(in_shift + 4 * t.q) : f4zero();
  const int Cp = 4 * Cv, Ct = Co / 2;
  float4 wq[CO], dwacc[CO];
  float dbacc[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    wq[co] = f4zero();
    dwacc[co] = f4zero();
    dbacc[co] = 0.f;
    if (co < Co && t.active) {
      const int c0 = 4 * t.q;
      wq[co].x = c0 + 0 < C ? w[co * C + c0 + 0] : 0.f;
      wq[co].y = c0 + 1 < C ? w[co * C + c0 + 1] : 0.f;
      wq[co].z = c0 + 2 < C ? w[co * C + c0 + 2] : 0.f;
      wq[co].w = c0 + 3 < C ? w[co * C + c0 + 3] : 0.f;
    }
  }
  const float coef = dloss ? dloss[s] * inv_count : 0.f;
  if (t.active) {
    const int P = N * HW;
    for (int p0 = t.p; p0 < P; p0 += UN * t.pstep) {
      float dl[UN][CO], mu[UN][CO / 2], lp[UN][CO / 2], y[UN][CO / 2], mk[UN];
      float4 av[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int p = min(p0 + u * t.pstep, P - 1);  // past the end: a valid pixel, results discarded
        const int n = p / HW;
        const int yx = p - n * HW;
        const int64_t obase = (((int64_t)n * S + s) * Co) * HW + yx;
#pragma unroll
        for (int co = 0; co < CO; ++co) dl[u][co] = (dout && co < Co) ? dout[obase + (int64_t)co * HW] : 0.f;
        mk[u] = 1.f;
        if (dloss) {
          const int64_t src = perm ? perm[(int64_t)s * N + n] : n;
          if (mask) mk[u] = mask[src * HW + yx];
#pragma unroll
          for (int tc = 0; tc < CO / 2; ++tc)
            if (tc < Ct) {
              mu[u][tc] = out[obase + (int64_t)tc * HW];
              lp[u][tc] = out[obase + (int64_t)(Ct + tc) * HW];
              y[u][tc] = label[(src * Ct + tc) * HW + yx];
            }
        }
        av[u] = ld4(a + (size_t)p * lda + 4 * t.q);
        if (fin) av[u] = bn_relu4(av[u], isc, ish);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int p = p0 + u * t.pstep;
        if (p >= P) break;
        if (dloss) {
#pragma unroll
          for (int tc = 0; tc < CO / 2; ++tc)
            if (tc < Ct) {
              float gm, gl;
              nll_grad(kind, mu[u][tc] - y[u][tc], lp[u][tc], eps_min, eps_max, &gm, &gl);
#pragma unroll
              for (int co = 0; co < CO; ++co) {  // dl[tc] and dl[Ct + tc], with compile-time register indices
                if (co == tc) dl[u][co] += coef * mk[u] * gm;
                if (co == Ct + tc) dl[u][co] += coef * mk[u] * gl;
              }
            }
        }
        float4 g = f4zero();
#pragma unroll
        for (int co = 0; co < CO; ++co)
          if (co < Co) {
            g.x = fmaf(dl[u][co], wq[co].x, g.x);
            g.y = fmaf(dl[u][co], wq[co].y, g.y);
            g.z = fmaf(dl[u][co], wq[co].z, g.z);
            g.w = fmaf(dl[u][co], wq[co].w, g.w);
            dwacc[co].x = fmaf(dl[u][co], av[u].x, dwacc[co].x);
            dwacc[co].y = fmaf(dl[u][co], av[u].y, dwacc[co].y);
            dwacc[co].z = fmaf(dl[u][co], av[u].z, dwacc[co].z);
            dwacc[co].w = fmaf(dl[u][co], av[u].w, dwacc[co].w);
            dbacc[co] += dl[u][co];
          }
        st4(da + (size_t)p * Cp + 4 * t.q, g);
      }
    }
  }
  // partial row: [Co][Cp] weight gradient followed by [Co] bias gradient (from quad 0 threads)
  float* row = partial + (size_t)blockIdx.x * (Co * Cp + Co);
  for (int co = 0; co < Co; ++co) {
    float4 v = f4zero();
    float b = 0.f;
#pragma unroll
    for (int k = 0; k < CO; ++k)
      if (k == co) {
        v = dwacc[k];
        b = dbacc[k];
      }
    const float4 sw = quad_block_sum(v, t, red);
    if (t.pl == 0 && t.q < Cv) st4(row + co * Cp + 4 * t.q, sw);
    const float4 sb = quad_block_sum(make_float4(b, 0.f, 0.f, 0.f), t, red);
    if (t.pl == 0 && t.q == 0) row[Co * Cp + co] = sb.x;
  }
}

int head_bwd_launch(const void* a, int dt, int lda, const float* w, int C, int Cp, int Co, int N, int S, int s, int HW,
                    const float* out, const float* dout, const float* dloss, const float* label, const float* mask,
                    const int64_t* perm, int kind, float eps_min, float eps_max, void* da, float* partial, int* rows,
                    hipStream_t st, const float* in_scale, const float* in_shift) {
  if (Co > kMaxHeadOut || (Co & 1)) {
    set_error("head: out_channels %d unsupported", Co);
    return MIMO_ERR_INVALID;
  }
  const int Cv = Cp / 4;
  dim3 grid = pq_grid(Cv, (int64_t)N * HW);
  grid.y = 1;  // Cv <= 64 for the head (filter_base_count <= 256)
  *rows = grid.x;
  const float inv_count = 1.f / (float)((double)N * (Co / 2) * HW);
#define HEAD_BWD_LAUNCH(CC)                                                                                              \
  MIMO_ST_DISPATCH(dt, T, hipLaunchKernelGGL((head_bwd_kernel<CC, T>), grid, dim3(256), 0, st, (const T*)a, lda, w, C, Cv, Co, N, S, s, \
                                             HW, out, dout, dloss, label, mask, perm, kind, eps_min, eps_max, inv_count, (T*)da,  \
                                             partial, in_scale, in_shift))
  if (Co <= 2) {
    HEAD_BWD_LAUNCH(2);
  } else if (Co <= 4) {
    HEAD_BWD_LAUNCH(4);
  } else {
    HEAD_BWD_LAUNCH(kMaxHeadOut);
  }
#undef HEAD_BWD_LAUNCH
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

__global__ void head_bwd_finalize_kernel(const double* __restrict__ sums, int chunks, int C, int Cp, int Co,
                                         float* __restrict__ dw, float* __restrict__ db) {
  const int cols = Co * Cp + Co;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cols) return;
  double s = 0.0;
  for (int k = 0; k < chunks; ++k) s += sums[(size_t)k * cols + i];
  if (i < Co * Cp) {
    const int co = i / Cp, c = i - co * Cp;
    if (c < C) dw[co * C + c] = (float)s;
  } else {
    db[i - Co * Cp] = (float)s;
  }
}

// rowsum + head_bwd_finalize in one launch: partial rows [rows][Co*Cp + Co] from head_bwd
__global__ __launch_bounds__(kColsumThreads) void head_bwd_stats_kernel(const float* __restrict__ partial, int rows, int C,
                                                                        int Cp, int Co, float* __restrict__ dw,
                                                                        float* __restrict__ db, double* __restrict__ scratch,
                                                                        int scratch_cols, int* __restrict__ tickets) {
  __shared__ double red[2048];
  const int cols = Co * Cp + Co;
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  double s, unused;
  if (!grid_colsum2(partial, rows, (size_t)cols, i, i, i < cols, false, red, scratch, scratch_cols, tickets, &s, &unused)) return;
  if (i >= cols || threadIdx.x >= 64) return;
  if (i < Co * Cp) {
    const int co = i / Cp, c = i - co * Cp;
    if (c < C) dw[co * C + c] = (float)s;
  } else {
    db[i - Co * Cp] = (float)s;
  }
}

int head_bwd_stats_launch(const float* partial, int rows, int C, int Cp, int Co, float* dw, float* db, const ColsumScratch& cs,
                          hipStream_t st) {
  const int groups = ceil_div(Co * Cp + Co, 64);
  hipLaunchKernelGGL(head_bwd_stats_kernel, dim3(groups, colsum_chunks(rows)), dim3(kColsumThreads), 0, st, partial, rows, C, Cp,
                     Co, dw, db, cs.sums, groups * 64, cs.tickets);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

int head_bwd_finalize_launch(const double* sums, int chunks, int C, int Cp, int Co, float* dw, float* db, hipStream_t st) {
  const int cols = Co * Cp + Co;
  hipLaunchKernelGGL(head_bwd_finalize_kernel, dim3(ceil_div(cols, 64)), dim3(64), 0, st, sums, chunks, C, Cp, Co, dw, db);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

}  // namespace mimo
