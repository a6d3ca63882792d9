// 3x3 convolution kernels for gfx950 (MI355X): forward / data-gradient as an LDS-tiled
// implicit GEMM on the f32-input MFMA (v_mfma_f32_16x16x4_f32, exact fp32 fma chain), and the
// weight gradient as a pixel-reduction GEMM on the same instruction.
//
// Replaces (reference, relative to /root/reference):
//   nn.Conv2d(k=3, padding=1, padding_mode="reflect") forward + autograd
//   mimo/models/mimo_components/components.py:23,26
//
// Layout: activations NHWC with the channel count padded to a multiple of 8 (zero-filled).
// Implicit GEMM: M = output pixels of a TR x TC tile (linearised, 16 per MFMA fragment),
// N = output channels (16 per fragment), K = 9 taps x input channels (4 per MFMA).
#include <algorithm>

#include <hip/hip_ext.h>

#include "common.h"

namespace mimo {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxTilePix = 360;  // (TR+2)*(TC+2) upper bound held in LDS

// ---------------------------------------------------------------------------------------
// tile geometry (host)
// ---------------------------------------------------------------------------------------
static void pick_tile(int Ho, int Wo, int* TR, int* TC) { sched::pick_tile_n(Ho, Wo, 256, kMaxTilePix, TR, TC); }

int conv3x3_stat_rows(int N, int Ho, int Wo) {
  int TR, TC;
  pick_tile(Ho, Wo, &TR, &TC);
  return N * ceil_div(Ho, TR) * ceil_div(Wo, TC);
}

int conv3x3_pick_nfrag(int cout) { return sched::conv_pick_nfrag(cout); }
int conv3x3_cout_pad(int cout) { return sched::conv_cout_pad(cout); }

// ---------------------------------------------------------------------------------------
// forward-type kernel
// ---------------------------------------------------------------------------------------
template <int NFRAG, int CK>
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(ConvLaunch a, int TR, int TC, int tilesY, int tilesX) {
  constexpr int NB = NFRAG * 16;
  constexpr int XS = CK + 1;  // odd LDS pitch: conflict-free b32 fragment reads
  constexpr int Q = CK / 4;
  __shared__ float xs[kMaxTilePix * XS];
  __shared__ float ws[9 * NB * XS];
  __shared__ int goff[kMaxTilePix];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int TCP = TC + 2, TRP = TR + 2;
  const int npix_lds = TRP * TCP;
  const int npix_out = TR * TC;

  int bx = blockIdx.x;
  const int tx = bx % tilesX;
  bx /= tilesX;
  const int ty = bx % tilesY;
  const int n = bx / tilesY;
  const int y0 = ty * TR, x0 = tx * TC;
  const int co0 = blockIdx.y * NB;

  // global offset (floats, relative to image n) of every LDS tile pixel; -1 = zero fill
  for (int p = tid; p < npix_lds; p += 256) {
    const int tr = p / TCP, tc = p - tr * TCP;
    int iy = y0 - a.off + tr, ix = x0 - a.off + tc;
    int o;
    if (a.off == 1) {  // reflect (then clamp for tile overhang; overhang feeds masked outputs only)
      iy = iy < 0 ? -iy : iy;
      iy = iy >= a.Hi ? 2 * a.Hi - 2 - iy : iy;
      ix = ix < 0 ? -ix : ix;
      ix = ix >= a.Wi ? 2 * a.Wi - 2 - ix : ix;
      iy = min(max(iy, 0), a.Hi - 1);
      ix = min(max(ix, 0), a.Wi - 1);
      o = (iy * a.Wi + ix) * a.ldx;
    } else {
      o = (iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi) ? (iy * a.Wi + ix) * a.ldx : -1;
    }
    goff[p] = o;
  }

  const float* ximg = a.x + (size_t)n * a.Hi * a.Wi * a.ldx;

  int pbase[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    int idx = (wave * 4 + m) * 16 + lr;
    if (idx >= npix_out) idx = 0;
    const int r = idx / TC, c = idx - r * TC;
    pbase[m] = (r * TCP + c) * XS + g;
  }
  const int wbase = lr * XS + g;

  f32x4 acc[4][NFRAG];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int nf = 0; nf < NFRAG; ++nf) acc[m][nf] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int c0 = 0; c0 < a.cin_p; c0 += CK) {
    __syncthreads();  // goff ready (first pass) / previous chunk fully consumed
    for (int i = tid; i < npix_lds * Q; i += 256) {
      const int p = i / Q, q = i - p * Q;
      const int o = goff[p];
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (o >= 0) v = *reinterpret_cast<const float4*>(ximg + o + c0 + 4 * q);
      float* d = xs + p * XS + 4 * q;
      d[0] = v.x;
      d[1] = v.y;
      d[2] = v.z;
      d[3] = v.w;
    }
    for (int i = tid; i < 9 * NB * Q; i += 256) {
      const int q = i % Q;
      const int row = i / Q;  // tap*NB + co
      const int tap = row / NB, co = row - tap * NB;
      const float4 v =
          *reinterpret_cast<const float4*>(a.w + ((size_t)tap * a.cout_pad + co0 + co) * a.cin_p + c0 + 4 * q);
      float* d = ws + row * XS + 4 * q;
      d[0] = v.x;
      d[1] = v.y;
      d[2] = v.z;
      d[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int toff = (kh * TCP + kw) * XS;
        const float* wt = ws + (kh * 3 + kw) * NB * XS + wbase;
#pragma unroll
        for (int ks = 0; ks < Q; ++ks) {
          float bf[NFRAG], af[4];
#pragma unroll
          for (int nf = 0; nf < NFRAG; ++nf) bf[nf] = wt[nf * 16 * XS + ks * 4];
#pragma unroll
          for (int m = 0; m < 4; ++m) af[m] = xs[pbase[m] + toff + ks * 4];
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int nf = 0; nf < NFRAG; ++nf)
              acc[m][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[nf], acc[m][nf], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue: bias, store, per-channel sum / sum-of-squares for BatchNorm -------------
  float bv[NFRAG], s1[NFRAG], s2[NFRAG], esc[NFRAG], esh[NFRAG], emk[NFRAG];
#pragma unroll
  for (int nf = 0; nf < NFRAG; ++nf) {
    bv[nf] = a.bias ? a.bias[co0 + nf * 16 + lr] : 0.f;
    s1[nf] = 0.f;
    s2[nf] = 0.f;
    esc[nf] = a.ep_scale ? a.ep_scale[co0 + nf * 16 + lr] : 1.f;
    esh[nf] = a.ep_scale ? a.ep_shift[co0 + nf * 16 + lr] : 0.f;
    emk[nf] = (a.ep_mask && co0 + nf * 16 + lr < a.ep_mask_ld) ? a.ep_mask[(size_t)n * a.ep_mask_ld + co0 + nf * 16 + lr] : 1.f;
  }
  float* yimg = a.y + (size_t)n * a.Ho * a.Wo * a.ldy;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int idx = (wave * 4 + m) * 16 + g * 4 + r4;
      const int orow = idx / TC, ocol = idx - orow * TC;
      const int oy = y0 + orow, ox = x0 + ocol;
      const bool ok = idx < npix_out && oy < a.Ho && ox < a.Wo;
      if (ok) {
        float* yp = yimg + ((size_t)oy * a.Wo + ox) * a.ldy + co0 + lr;
#pragma unroll
        for (int nf = 0; nf < NFRAG; ++nf) {
          float v = acc[m][nf][r4] + bv[nf];
          if (a.ep_scale) {
            if (a.status && !isfinite(v)) atomicOr(a.status, 1);  // the ReLU below would drop a NaN
            v = fmaxf(fmaf(v, esc[nf], esh[nf]), 0.f) * emk[nf];
          }
          if (co0 + nf * 16 + lr < a.cout_store) yp[nf * 16] = v;
          s1[nf] += v;
          s2[nf] += v * v;
        }
      }
    }
  }
  if (a.stats) {
#pragma unroll
    for (int nf = 0; nf < NFRAG; ++nf) {
      s1[nf] += __shfl_xor(s1[nf], 16);
      s1[nf] += __shfl_xor(s1[nf], 32);
      s2[nf] += __shfl_xor(s2[nf], 16);
      s2[nf] += __shfl_xor(s2[nf], 32);
    }
    __syncthreads();  // all waves are done with ws
    float* red = ws;  // [4 waves][2][NB]
    if (g == 0) {
#pragma unroll
      for (int nf = 0; nf < NFRAG; ++nf) {
        red[(wave * 2 + 0) * NB + nf * 16 + lr] = s1[nf];
        red[(wave * 2 + 1) * NB + nf * 16 + lr] = s2[nf];
      }
    }
    __syncthreads();
    if (tid < 2 * NB) {
      const int which = tid / NB, c = tid - which * NB;
      const float v = red[(0 * 2 + which) * NB + c] + red[(1 * 2 + which) * NB + c] +
                      red[(2 * 2 + which) * NB + c] + red[(3 * 2 + which) * NB + c];
      a.stats[((size_t)blockIdx.x * 2 + which) * a.cout_pad + co0 + c] = v;
    }
  }
}

template <int NFRAG, int CK>
static int launch_conv(const ConvLaunch& a, int TR, int TC, int tilesY, int tilesX, hipStream_t stream) {
  dim3 grid(a.N * tilesY * tilesX, a.cout_pad / (NFRAG * 16));
  hipLaunchKernelGGL((conv3x3_mfma_kernel<NFRAG, CK>), grid, dim3(256), 0, stream, a, TR, TC, tilesY, tilesX);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

template <int NFRAG>
static int launch_conv_ck(const ConvLaunch& a, int TR, int TC, int tilesY, int tilesX, hipStream_t stream) {
  if (a.cin_p % 16 == 0) return launch_conv<NFRAG, 16>(a, TR, TC, tilesY, tilesX, stream);
  if (a.cin_p % 8 == 0) return launch_conv<NFRAG, 8>(a, TR, TC, tilesY, tilesX, stream);
  return launch_conv<NFRAG, 4>(a, TR, TC, tilesY, tilesX, stream);
}

int conv3x3_launch(const ConvLaunch& a, int* rows, hipStream_t stream) {
  if (a.cin_p % 4 != 0 || a.ldx % 4 != 0 || a.cout_pad % 16 != 0 || a.Hi < 2 || a.Wi < 2) {
    set_error("conv3x3: bad geometry cin_p=%d ldx=%d cout_pad=%d H=%d W=%d", a.cin_p, a.ldx, a.cout_pad, a.Hi, a.Wi);
    return MIMO_ERR_INVALID;
  }
  int TR, TC;
  pick_tile(a.Ho, a.Wo, &TR, &TC);
  const int tilesY = ceil_div(a.Ho, TR), tilesX = ceil_div(a.Wo, TC);
  if (rows) *rows = a.N * tilesY * tilesX;
  const int nfr = a.cout_pad / 16;
  int nfrag = 4;
  while (nfr % nfrag != 0) --nfrag;
  switch (nfrag) {
    case 4: return launch_conv_ck<4>(a, TR, TC, tilesY, tilesX, stream);
    case 3: return launch_conv_ck<3>(a, TR, TC, tilesY, tilesX, stream);
    case 2: return launch_conv_ck<2>(a, TR, TC, tilesY, tilesX, stream);
    default: return launch_conv_ck<1>(a, TR, TC, tilesY, tilesX, stream);
  }
}

// ---------------------------------------------------------------------------------------
// weight gradient: dW[tap][ci][co] = sum_pixels A(pixel+tap, ci) * dz(pixel, co)
// MFMA: rows = 16 input channels, cols = 16 output channels, K = 4 consecutive pixels.
// LDS holds both tiles channel-major with pitch == 2 (mod 32): conflict-free b32 reads.
// ---------------------------------------------------------------------------------------
template <int TC>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgradLaunch a, int tilesY, int tilesX, int numTiles) {
  constexpr int TR = 256 / TC;
  constexpr int TCP = TC + 2, TRP = TR + 2;
  constexpr int APIX = TRP * TCP;
  constexpr int APITCH = ((APIX + 31) / 32) * 32 + 2;
  constexpr int DPITCH = 258;
  __shared__ float as[32 * APITCH];
  __shared__ float ds[32 * DPITCH];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int mi = wave >> 1, ni = wave & 1;
  const int coTiles = a.cout_pad / 32;
  const int ciT = blockIdx.x / coTiles, coT = blockIdx.x - ciT * coTiles;
  const int ci0 = ciT * 32, co0 = coT * 32;

  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int abase = (mi * 16 + lr) * APITCH + g;
  const int bbase = (ni * 16 + lr) * DPITCH + g;

  for (int tile = blockIdx.y; tile < numTiles; tile += gridDim.y) {
    int t = tile;
    const int tx = t % tilesX;
    t /= tilesX;
    const int ty = t % tilesY;
    const int n = t / tilesY;
    const int y0 = ty * TR, x0 = tx * TC;
    const float* ximg = a.x + (size_t)n * a.H * a.W * a.ldx;
    const float* dimg = a.dz + (size_t)n * a.H * a.W * a.lddz;
    __syncthreads();
    for (int i = tid; i < APIX * 8; i += 256) {
      const int p = i >> 3, q = i & 7;
      const int tr = p / TCP, tc = p - tr * TCP;
      int iy = y0 - 1 + tr, ix = x0 - 1 + tc;
      iy = iy < 0 ? -iy : iy;
      iy = iy >= a.H ? 2 * a.H - 2 - iy : iy;
      ix = ix < 0 ? -ix : ix;
      ix = ix >= a.W ? 2 * a.W - 2 - ix : ix;
      iy = min(max(iy, 0), a.H - 1);
      ix = min(max(ix, 0), a.W - 1);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ci0 + 4 * q < a.cin_p) v = *reinterpret_cast<const float4*>(ximg + ((size_t)iy * a.W + ix) * a.ldx + ci0 + 4 * q);
      float* d = as + (4 * q) * APITCH + p;
      d[0] = v.x;
      d[APITCH] = v.y;
      d[2 * APITCH] = v.z;
      d[3 * APITCH] = v.w;
    }
    for (int i = tid; i < 256 * 8; i += 256) {
      const int p = i >> 3, q = i & 7;
      const int r = p / TC, c = p - r * TC;
      const int y = y0 + r, x = x0 + c;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (y < a.H && x < a.W && co0 + 4 * q < a.cout_p)
        v = *reinterpret_cast<const float4*>(dimg + ((size_t)y * a.W + x) * a.lddz + co0 + 4 * q);
      float* d = ds + (4 * q) * DPITCH + p;
      d[0] = v.x;
      d[DPITCH] = v.y;
      d[2 * DPITCH] = v.z;
      d[3 * DPITCH] = v.w;
    }
    __syncthreads();
#pragma unroll 4
    for (int ks = 0; ks < 64; ++ks) {
      const int pix = ks * 4;
      const int r = pix / TC, c = pix - r * TC;
      const float b = ds[bbase + pix];
      const float* ap = as + abase + r * TCP + c;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kh * TCP + kw], b, acc[kh * 3 + kw], 0, 0, 0);
    }
  }
  float* out = a.partial + (size_t)blockIdx.y * 9 * a.cin_pad * a.cout_pad;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int ci = ci0 + mi * 16 + g * 4 + r4;
      const int co = co0 + ni * 16 + lr;
      out[((size_t)t * a.cin_pad + ci) * a.cout_pad + co] = acc[t][r4];
    }
}

static int wgrad_tc(int W) { return W >= 24 ? 32 : (W >= 12 ? 16 : 8); }

static int wgrad_num_tiles(int N, int H, int W) {
  const int tc = wgrad_tc(W), tr = 256 / tc;
  return N * ceil_div(H, tr) * ceil_div(W, tc);
}

int wgrad_pick_splits(int N, int H, int W, int cin_pad, int cout_pad) {
  const int wtiles = (cin_pad / 32) * (cout_pad / 32);
  const int tiles = wgrad_num_tiles(N, H, W);
  int splits = ceil_div(2048, wtiles);
  if (splits > tiles) splits = tiles;
  if (splits > 1024) splits = 1024;
  if (splits < 1) splits = 1;
  return splits;
}

int wgrad_launch(const WgradLaunch& a, hipStream_t stream) {
  if (a.cin_pad % 32 || a.cout_pad % 32 || a.cin_p % 4 || a.cout_p % 4 || a.ldx % 4 || a.lddz % 4 || a.H < 2 || a.W < 2) {
    set_error("wgrad: bad geometry");
    return MIMO_ERR_INVALID;
  }
  const int tc = wgrad_tc(a.W), tr = 256 / tc;
  const int tilesY = ceil_div(a.H, tr), tilesX = ceil_div(a.W, tc);
  const int numTiles = a.N * tilesY * tilesX;
  dim3 grid((a.cin_pad / 32) * (a.cout_pad / 32), a.splits);
  if (tc == 32)
    hipLaunchKernelGGL((wgrad_mfma_kernel<32>), grid, dim3(256), 0, stream, a, tilesY, tilesX, numTiles);
  else if (tc == 16)
    hipLaunchKernelGGL((wgrad_mfma_kernel<16>), grid, dim3(256), 0, stream, a, tilesY, tilesX, numTiles);
  else
    hipLaunchKernelGGL((wgrad_mfma_kernel<8>), grid, dim3(256), 0, stream, a, tilesY, tilesX, numTiles);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// dw (torch OIHW [cout][cin][3][3]) <- sum over splits of partial[split][tap][ci_pad][co_pad];
// cin_map[padded ci] = logical ci or -1.  Two stages, both fully coalesced:
//   A (only when splits > kReduceFan): slab group sums, element-wise, grid over (elements, groups)
//   B: per (32 ci x 32 co) tile, sum <= kReduceFan slabs with co-contiguous reads, transpose the
//      [9][32][32] tile through LDS and write torch's layout as 288-float contiguous runs per co
constexpr int kReduceFan = 16;  // slabs summed per group (stage A)

__global__ void wgrad_group_sum_kernel(const float* __restrict__ partial, int splits, size_t slab, float* __restrict__ out) {
  const int grp = blockIdx.y;
  const int s0 = grp * kReduceFan, s1 = min(splits, s0 + kReduceFan);
  const size_t n4 = slab / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = s0; s < s1; ++s) {
      const float4 v = reinterpret_cast<const float4*>(partial + (size_t)s * slab)[i];
      acc.x += v.x;
      acc.y += v.y;
      acc.z += v.z;
      acc.w += v.w;
    }
    reinterpret_cast<float4*>(out + (size_t)grp * slab)[i] = acc;
  }
}

__device__ __forceinline__ void wgrad_reduce_tile(const float* __restrict__ partial, int splits, int cin_pad, int cout_pad,
                                                  const int* __restrict__ cin_map, int cin_p, int cin, int cout,
                                                  float* __restrict__ dw, int block, float mul) {
  // 8 output channels per pass: a 9 KB LDS tile, so the kernel fits next to the 145-159 KB workgroups of
  // the persistent convolution kernels it runs beside (side stream) instead of waiting for their CUs
  constexpr int PITCH = 289;  // 32*9 + 1
  constexpr int COB = 8;
  __shared__ float tile[COB * PITCH];
  // one (32 ci x 8 co) sub-tile per workgroup: the four sequential passes of a 32 x 32 tile were four exposed
  // load -> LDS -> store round trips (13 us even for a 36 KB slab)
  const int coTiles = (cout_pad + COB - 1) / COB;  // cin_pad / cout_pad are multiples of 16
  const int ciT = block / coTiles, coT = block - ciT * coTiles;
  const int ci0 = ciT * 32;
  const size_t slab4 = (size_t)9 * cin_pad * cout_pad / 4;
  {
    const int co0 = coT * COB;
    // the [9][32 ci][8 co] sub-tile = 576 float4 units, up to 3 per thread: independent accumulators,
    // so the loads of all units and slabs are in flight together (the kernel is pure latency otherwise)
    float4 acc[3];
    const float4* src[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int u = threadIdx.x + k * 256;
      const int row = u >> 1, c4 = u & 1;  // row = tap*32 + ci_local
      const int tap = row >> 5, cil = row & 31;
      const bool ok = u < 576 && ci0 + cil < cin_pad && co0 + 4 * c4 < cout_pad;
      src[k] = ok ? reinterpret_cast<const float4*>(partial + ((size_t)tap * cin_pad + ci0 + cil) * cout_pad + co0 + 4 * c4)
                  : nullptr;
      acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int sidx = 0; sidx < splits; ++sidx) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (src[k]) {
          const float4 v = src[k][(size_t)sidx * slab4];
          acc[k].x += v.x;
          acc[k].y += v.y;
          acc[k].z += v.z;
          acc[k].w += v.w;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int u = threadIdx.x + k * 256;
      if (u < 576) {
        const int row = u >> 1, c4 = u & 1;
        const int tap = row >> 5, cil = row & 31;
        float* t = tile + (4 * c4) * PITCH + cil * 9 + tap;
        t[0] = acc[k].x;
        t[PITCH] = acc[k].y;
        t[2 * PITCH] = acc[k].z;
        t[3 * PITCH] = acc[k].w;
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < COB * 288; i += 256) {
      const int co = i / 288, j = i - co * 288;
      const int cil = j / 9, tap = j - cil * 9;
      const int cip = ci0 + cil;
      if (co0 + co >= cout || cip >= cin_p) continue;
      const int ci = cin_map ? cin_map[cip] : (cip < cin ? cip : -1);
      if (ci < 0) continue;
      dw[((size_t)(co0 + co) * cin + ci) * 9 + tap] = tile[co * PITCH + j] * mul;  // (mul: a power of two or 1, exact)
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int splits, int cin_pad,
                                                           int cout_pad, const int* __restrict__ cin_map, int cin_p,
                                                           int cin, int cout, float* __restrict__ dw,
                                                           const float* __restrict__ dz_absmax, int dz_absmax_n) {
  wgrad_reduce_tile(partial, splits, cin_pad, cout_pad, cin_map, cin_p, cin, cout, dw, (int)blockIdx.x,
                    dz_absmax ? wg_dz_scale(__float_as_uint(wg_dz_absmax(dz_absmax, dz_absmax_n)), true) : 1.f);
}

// scratch (floats) the reduction needs behind the `splits` slabs of the wgrad kernels
size_t wgrad_reduce_scratch(int splits, int cin_pad, int cout_pad) {
  size_t slabs = 0;
  for (int n = splits; n > 1;) {  // every stage-A output (upper bound: all stages down to one slab)
    n = ceil_div(n, kReduceFan);
    slabs += n;
  }
  return slabs * 9 * cin_pad * cout_pad;
}

int wgrad_reduce_launch(const float* partial, int splits, int cin_pad, int cout_pad, const int* cin_map, int cin_p,
                        int cin, int cout, float* dw, hipStream_t stream, const float* dz_absmax, int dz_absmax_n, hipEvent_t done) {
  const size_t slab = (size_t)9 * cin_pad * cout_pad;
  const float* src = partial;
  int n = splits;
  const int blocks = ceil_div(cin_pad, 32) * ceil_div(cout_pad, 8);
  // few output tiles (small layers, hundreds of splits): the transposing kernel has too few workgroups to
  // stream many slabs, so stage A sums all the way down to one slab
  const int final_fan = blocks >= 512 ? kReduceFan : 1;
  while (n > final_fan) {  // stage A (repeated for large split counts); output behind the inputs
    const int groups = ceil_div(n, kReduceFan);
    float* out = const_cast<float*>(src) + (size_t)n * slab;
    const int bx = (int)std::min<size_t>((slab / 4 + 255) / 256, 512);
    hipLaunchKernelGGL(wgrad_group_sum_kernel, dim3(bx, groups), dim3(256), 0, stream, src, n, slab, out);
    MIMO_KERNEL_CHECK();
    src = out;
    n = groups;
  }
  // (`done`: recorded when the last kernel of the reduction completes, attached to its launch — elementwise.hip bn_bwd_apply_launch)
  hipExtLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, stream, nullptr, done, 0, src, n, cin_pad, cout_pad,
                        cin_map, cin_p, cin, cout, dw, dz_absmax, dz_absmax_n);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// ---------------------------------------------------------------------------------------
// weight packing: torch OIHW -> dst[tap][row][col]
//   transposed == 0 (forward):  row -> co = row_map[row], col -> ci = col_map[col], tap = kh*3+kw
//   transposed == 1 (dgrad):    row -> ci = row_map[row], col -> co = col_map[col], tap = (2-kh)*3+(2-kw)
// map value -1 = zero padding.
// ---------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ dst, int cout, int cin,
                                    int rows_pad, int cols, const int* __restrict__ row_map,
                                    const int* __restrict__ col_map, int transposed) {
  const int total = 9 * rows_pad * cols;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int col = i % cols;
    const int rest = i / cols;
    const int row = rest % rows_pad, tap = rest / rows_pad;
    const int rm = row_map[row], cm = col_map[col];
    float v = 0.f;
    if (rm >= 0 && cm >= 0) {
      const int co = transposed ? cm : rm, ci = transposed ? rm : cm;
      const int kh = transposed ? 2 - tap / 3 : tap / 3, kw = transposed ? 2 - tap % 3 : tap % 3;
      v = w[(((size_t)co * cin + ci) * 3 + kh) * 3 + kw];
    }
    dst[i] = v;
  }
}

int pack_weights_launch(const float* w, float* dst, int cout, int cin, int rows_pad, int cols, const int* row_map,
                        const int* col_map, int transposed, hipStream_t stream) {
  const int total = 9 * rows_pad * cols;
  const int blocks = min(ceil_div(total, 256), 4096);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, stream, w, dst, cout, cin, rows_pad, cols,
                     row_map, col_map, transposed);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

}  // namespace mimo
