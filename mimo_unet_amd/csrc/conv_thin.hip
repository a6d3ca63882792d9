// The image convolution of an encoder (2..4 input channels -> filter_base_count) as plain fp32 FMA kernels for gfx950.
//
// Replaces (reference, relative to /root/reference): the FIRST nn.Conv2d(k=3, padding=1, padding_mode="reflect") of
// encoder.in_convs[s] — forward and weight gradient — mimo/models/mimo_components/components.py:23 on the widths of
// mimo/models/mimo_components/model.py:150-160.
//
// With <= 4 input channels the layer is 18..36 multiply-adds per output value next to 4 bytes written (forward) or read
// (weight gradient): it is bound by the HBM pass over the 30-channel tensor, not by arithmetic, and padding its K
// dimension to an MFMA chunk (16 / 32 channels) only added staging and zero products in front of that pass.  One thread
// owns one pixel x four output channels (the mapping of the bandwidth kernels in elementwise.hip): its 9 x CIN x 4 weights
// (forward) or weight-gradient accumulators (backward) live in registers, the 3x3 neighbourhoods come from a
// reflect-padded halo tile of the image in LDS (loaded once per 8 x 32-pixel tile, double-buffered), the 30-channel tensor
// is touched once with 16-byte accesses.  (A first version read the neighbourhood straight from global memory: nine loads
// per thread bound it on the texture-address path, slower than the MFMA kernels — profiles/r04/bwd_source_fusion.txt.)
// Arithmetic: exact fp32 products, fp32 accumulate (the arithmetic of the fp32 kernel family this layer ran on before).
#include <algorithm>

#include "common.h"

namespace mimo {

namespace {

__device__ __forceinline__ float4 ld4f(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4f(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// thread -> (pixel lane, channel quad): QB quads per pixel, 256 / QB pixels per workgroup and iteration
struct ThinMap {
  int q, pl, QB, PPI;
  bool active;  // (256 % Cv != 0: the last threads of the workgroup own no pixel)
};
__device__ __forceinline__ ThinMap thin_map(int Cv) {
  ThinMap t;
  t.QB = Cv;  // Cv <= 64 (checked by the launchers)
  t.PPI = 256 / Cv;
  t.q = threadIdx.x % Cv;
  t.pl = threadIdx.x / Cv;
  t.active = t.pl < t.PPI;
  return t;
}

// A workgroup walks 8 x 32-pixel tiles (persistent, grid-stride); the tile's reflect-padded 10 x 34 halo of the image sits
// in LDS (CINP floats per pixel), double-buffered: the halo of the next tile is loaded into registers in front of the
// arithmetic of the current one and stored behind it — one barrier per tile.  The threads of a pixel read the same LDS
// addresses (broadcast).
constexpr int kTH = 8, kTW = 32, kHaloW = kTW + 2, kHaloPix = (kTH + 2) * kHaloW;  // 340 halo pixels, 256 outputs
constexpr int kHaloPerThread = (kHaloPix + 255) / 256;                             // 2

template <int CIN>
struct Px {  // one pixel's input channels as stored in LDS
  static constexpr int N = CIN == 3 ? 4 : CIN;
  float v[N];
};
template <int CIN>
__device__ __forceinline__ Px<CIN> load_px(const float* p) {
  Px<CIN> r;
  if constexpr (CIN == 1) {
    r.v[0] = p[0];
  } else if constexpr (CIN == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    r.v[0] = t.x;
    r.v[1] = t.y;
  } else {
    const float4 t = ld4f(p);
    r.v[0] = t.x;
    r.v[1] = t.y;
    r.v[2] = t.z;
    r.v[3] = t.w;
  }
  return r;
}
template <int CIN>
__device__ __forceinline__ void store_px(float* p, const Px<CIN>& r) {
  if constexpr (CIN == 1) {
    p[0] = r.v[0];
  } else if constexpr (CIN == 2) {
    *reinterpret_cast<float2*>(p) = make_float2(r.v[0], r.v[1]);
  } else {
    st4f(p, make_float4(r.v[0], r.v[1], r.v[2], r.v[3]));
  }
}

struct ThinTile {
  int n, y0, x0;
};
__device__ __forceinline__ ThinTile thin_tile(int t, int tilesY, int tilesX) {
  ThinTile r;
  const int tx = t % tilesX;
  t /= tilesX;
  r.x0 = tx * kTW;
  r.y0 = (t % tilesY) * kTH;
  r.n = t / tilesY;
  return r;
}
// this thread's halo pixels of a tile, from the image (reflect padding; rows / columns past the image: clamped, they feed
// masked outputs only)
template <int CIN>
__device__ __forceinline__ void halo_load(const float* x, int ldx, int H, int W, const ThinTile& t, Px<CIN> pre[kHaloPerThread]) {
  const float* img = x + (size_t)t.n * H * W * ldx;
#pragma unroll
  for (int k = 0; k < kHaloPerThread; ++k) {
    const int hp = min((int)threadIdx.x + k * 256, kHaloPix - 1);
    const int hr = hp / kHaloW, hc = hp - hr * kHaloW;
    int y = t.y0 - 1 + hr, xx = t.x0 - 1 + hc;
    y = y < 0 ? -y : y;
    y = y >= H ? 2 * H - 2 - y : y;
    xx = xx < 0 ? -xx : xx;
    xx = xx >= W ? 2 * W - 2 - xx : xx;
    y = min(max(y, 0), H - 1);
    xx = min(max(xx, 0), W - 1);
    pre[k] = load_px<CIN>(img + ((size_t)y * W + xx) * ldx);
  }
}
template <int CIN>
__device__ __forceinline__ void halo_store(float* xs, const Px<CIN> pre[kHaloPerThread]) {
#pragma unroll
  for (int k = 0; k < kHaloPerThread; ++k) {
    const int hp = (int)threadIdx.x + k * 256;
    if (hp < kHaloPix) store_px<CIN>(xs + hp * Px<CIN>::N, pre[k]);
  }
}

// ---------------------------------------------------------------------------------------
// forward: y[p][co] = bias[co] + sum_{tap, ci} x[reflect(p + tap)][ci] * w[tap][co][ci]
// (ConvLaunch as conv3x3_launch takes it: packed weights [9][cout_pad][cin_p], the inference epilogue, the BatchNorm
// partial sums — one row per workgroup)
// ---------------------------------------------------------------------------------------
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_thin_fwd_kernel(ConvLaunch a, int Cv, int tilesY, int tilesX, int numTiles) {
  constexpr int CP = Px<CIN>::N;
  __shared__ __attribute__((aligned(16))) float xs[2][kHaloPix * CP];
  __shared__ float4 red[256];
  const ThinMap t = thin_map(Cv);
  const int c0 = 4 * t.q;
  float w[9][CIN][4];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
      for (int j = 0; j < 4; ++j) w[tap][ci][j] = a.w[((size_t)tap * a.cout_pad + c0 + j) * a.cin_p + ci];
  const float4 bv = a.bias ? ld4f(a.bias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 esc = a.ep_scale ? ld4f(a.ep_scale + c0) : make_float4(1.f, 1.f, 1.f, 1.f);
  const float4 esh = a.ep_scale ? ld4f(a.ep_shift + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  Px<CIN> pre[kHaloPerThread];
  int tile = blockIdx.x, buf = 0;
  if (tile < numTiles) {
    halo_load<CIN>(a.x, a.ldx, a.Hi, a.Wi, thin_tile(tile, tilesY, tilesX), pre);
    halo_store<CIN>(xs[0], pre);
  }
  __syncthreads();
  for (; tile < numTiles; tile += gridDim.x, buf ^= 1) {
    const ThinTile tt = thin_tile(tile, tilesY, tilesX);
    const int next = tile + gridDim.x;
    if (next < numTiles) halo_load<CIN>(a.x, a.ldx, a.Hi, a.Wi, thin_tile(next, tilesY, tilesX), pre);
    float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
    if (a.ep_scale && a.ep_mask) {
      const float* mp = a.ep_mask + (size_t)tt.n * a.ep_mask_ld;
      m.x = c0 + 0 < a.ep_mask_ld ? mp[c0 + 0] : 1.f;
      m.y = c0 + 1 < a.ep_mask_ld ? mp[c0 + 1] : 1.f;
      m.z = c0 + 2 < a.ep_mask_ld ? mp[c0 + 2] : 1.f;
      m.w = c0 + 3 < a.ep_mask_ld ? mp[c0 + 3] : 1.f;
    }
    const float* xb = xs[buf];
    float* yimg = a.y + (size_t)tt.n * a.Ho * a.Wo * a.ldy + c0;
    if (t.active)
      for (int i = t.pl; i < kTH * kTW; i += t.PPI) {
        const int r = i / kTW, c = i - r * kTW;
        const int oy = tt.y0 + r, ox = tt.x0 + c;
        if (oy >= a.Ho || ox >= a.Wo) continue;
        float acc[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const Px<CIN> xv = load_px<CIN>(xb + ((r + kh) * kHaloW + c + kw) * CP);
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[j] = fmaf(xv.v[ci], w[kh * 3 + kw][ci][j], acc[j]);
          }
        float4 v = make_float4(acc[0], acc[1], acc[2], acc[3]);
        if (a.ep_scale) {
          if (a.status && !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w))) atomicOr(a.status, 1);
          v.x = fmaxf(fmaf(v.x, esc.x, esh.x), 0.f) * m.x;
          v.y = fmaxf(fmaf(v.y, esc.y, esh.y), 0.f) * m.y;
          v.z = fmaxf(fmaf(v.z, esc.z, esh.z), 0.f) * m.z;
          v.w = fmaxf(fmaf(v.w, esc.w, esh.w), 0.f) * m.w;
        }
        st4f(yimg + ((size_t)oy * a.Wo + ox) * a.ldy, v);
        s1.x += v.x;
        s1.y += v.y;
        s1.z += v.z;
        s1.w += v.w;
        s2.x = fmaf(v.x, v.x, s2.x);
        s2.y = fmaf(v.y, v.y, s2.y);
        s2.z = fmaf(v.z, v.z, s2.z);
        s2.w = fmaf(v.w, v.w, s2.w);
      }
    if (next < numTiles) halo_store<CIN>(xs[buf ^ 1], pre);
    __syncthreads();
  }
  if (!a.stats) return;
  // one partial row per workgroup: [2][cout_pad], the pixel lanes summed in a fixed order
  float* row = a.stats + (size_t)blockIdx.x * 2 * a.cout_pad;
  for (int which = 0; which < 2; ++which) {
    __syncthreads();
    red[threadIdx.x] = which ? s2 : s1;
    __syncthreads();
    if (t.pl == 0) {
      float4 s = red[t.q];
      for (int j = 1; j < t.PPI; ++j) {
        const float4 o = red[j * t.QB + t.q];
        s.x += o.x;
        s.y += o.y;
        s.z += o.z;
        s.w += o.w;
      }
      st4f(row + which * a.cout_pad + c0, s);
    }
  }
  for (int c = 4 * Cv + threadIdx.x; c < a.cout_pad; c += 256) {  // padded columns of the row
    row[c] = 0.f;
    row[a.cout_pad + c] = 0.f;
  }
}

// ---------------------------------------------------------------------------------------
// weight gradient: dW[tap][ci][co] = sum_p x[reflect(p + tap)][ci] * dz[p][co]; one partial [9][CIN][4 Cv] per workgroup,
// summed by the reduction below into torch's OIHW layout
// ---------------------------------------------------------------------------------------
template <int CIN>
__global__ __launch_bounds__(256) void wgrad_thin_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ dz,
                                                         int lddz, int N, int H, int W, int Cv, int tilesY, int tilesX,
                                                         int numTiles, float* __restrict__ partial) {
  constexpr int CP = Px<CIN>::N;
  constexpr int kG = 4;  // dz values of a tile per thread fetched together (8: two waves per SIMD, 94 us; 4: three, 86 us; 2: four, 92 us)
  __shared__ __attribute__((aligned(16))) float xs[2][kHaloPix * CP];
  __shared__ float4 red[256];
  const ThinMap t = thin_map(Cv);
  const int c0 = 4 * t.q;
  float4 acc[9][CIN];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) acc[tap][ci] = make_float4(0.f, 0.f, 0.f, 0.f);
  Px<CIN> pre[kHaloPerThread];
  int tile = blockIdx.x, buf = 0;
  if (tile < numTiles) {
    halo_load<CIN>(x, ldx, H, W, thin_tile(tile, tilesY, tilesX), pre);
    halo_store<CIN>(xs[0], pre);
  }
  __syncthreads();
  for (; tile < numTiles; tile += gridDim.x, buf ^= 1) {
    const ThinTile tt = thin_tile(tile, tilesY, tilesX);
    const int next = tile + gridDim.x;
    if (next < numTiles) halo_load<CIN>(x, ldx, H, W, thin_tile(next, tilesY, tilesX), pre);
    const float* xb = xs[buf];
    const float* dimg = dz + (size_t)tt.n * H * W * lddz + c0;
    if (t.active)
      for (int i0 = t.pl; i0 < kTH * kTW; i0 += kG * t.PPI) {
        // (dz one tile ahead as well — 32 more registers — dropped the kernel to one wave per SIMD: 94 -> 118 us)
        float4 g[kG];
#pragma unroll
        for (int u = 0; u < kG; ++u) {  // a pixel outside the tile / image contributes zero
          const int i = i0 + u * t.PPI, r = i / kTW, c = i - r * kTW;
          const int oy = tt.y0 + r, ox = tt.x0 + c;
          g[u] = (i < kTH * kTW && oy < H && ox < W) ? ld4f(dimg + ((size_t)oy * W + ox) * lddz) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < kG; ++u) {
          const int i = min(i0 + u * t.PPI, kTH * kTW - 1), r = i / kTW, c = i - r * kTW;
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const Px<CIN> xv = load_px<CIN>(xb + ((r + kh) * kHaloW + c + kw) * CP);
#pragma unroll
              for (int ci = 0; ci < CIN; ++ci) {
                float4& d = acc[kh * 3 + kw][ci];
                d.x = fmaf(xv.v[ci], g[u].x, d.x);
                d.y = fmaf(xv.v[ci], g[u].y, d.y);
                d.z = fmaf(xv.v[ci], g[u].z, d.z);
                d.w = fmaf(xv.v[ci], g[u].w, d.w);
              }
            }
        }
      }
    if (next < numTiles) halo_store<CIN>(xs[buf ^ 1], pre);
    __syncthreads();
  }
  float* out = partial + (size_t)blockIdx.x * 9 * CIN * 4 * Cv;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      __syncthreads();
      red[threadIdx.x] = acc[tap][ci];
      __syncthreads();
      if (t.pl == 0) {
        float4 s = red[t.q];
        for (int j = 1; j < t.PPI; ++j) {
          const float4 o = red[j * t.QB + t.q];
          s.x += o.x;
          s.y += o.y;
          s.z += o.z;
          s.w += o.w;
        }
        st4f(out + ((size_t)tap * CIN + ci) * 4 * Cv + c0, s);
      }
    }
}

// dw[co][ci][tap] (torch OIHW) = sum over the workgroups' partials; one workgroup per (tap, ci), fixed summation order
__global__ __launch_bounds__(256) void wgrad_thin_reduce_kernel(const float* __restrict__ partial, int blocks, int cin, int Cp,
                                                                int cout, float* __restrict__ dw) {
  __shared__ double red[256];
  const int tc = blockIdx.x;  // tap * cin + ci
  const int tap = tc / cin, ci = tc - tap * cin;
  const int lanes = 256 / Cp;  // Cp <= 256: `lanes` partial rows in flight per channel
  const int c = threadIdx.x % Cp, l = threadIdx.x / Cp;
  double s = 0.0;
  if (l < lanes)
    for (int b = l; b < blocks; b += lanes) s += (double)partial[((size_t)b * 9 * cin + tc) * Cp + c];
  red[threadIdx.x] = s;
  __syncthreads();
  if (l == 0 && c < cout) {
    for (int j = 1; j < lanes; ++j) s += red[j * Cp + c];
    dw[((size_t)c * cin + ci) * 9 + tap] = (float)s;
  }
}

// MIMO_CONV_THIN: 1 (default) on, 0 off, 2 = the weight gradient too whatever the tile count (tests)
int thin_level() {
  static const int v = getenv("MIMO_CONV_THIN") ? atoi(getenv("MIMO_CONV_THIN")) : 1;
  return v;
}
bool thin_enabled() { return thin_level() != 0; }

}  // namespace

constexpr int kThinBlocks = 2048;  // partial-row bound of the one-launch BatchNorm statistics (and of the scratch)

// one full round of resident workgroups (the kernels hold 9 x CIN x 4 weights / accumulators per thread: 2..7 per CU)
template <typename K>
static int resident_blocks(K kernel) {
  int dev = 0, cus = 256, per_cu = 2;
  if (hipGetDevice(&dev) == hipSuccess) {
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
  }
  return std::min(cus * per_cu, kThinBlocks);
}
struct ThinCaps {
  int fwd[4], wg[4];
};
// (queried once, from conv3x3_thin_ok — i.e. at plan creation, never inside a stream capture)
static const ThinCaps& thin_caps() {
  static const ThinCaps c = {{resident_blocks(conv3x3_thin_fwd_kernel<1>), resident_blocks(conv3x3_thin_fwd_kernel<2>),
                              resident_blocks(conv3x3_thin_fwd_kernel<3>), resident_blocks(conv3x3_thin_fwd_kernel<4>)},
                             {resident_blocks(wgrad_thin_kernel<1>), resident_blocks(wgrad_thin_kernel<2>),
                              resident_blocks(wgrad_thin_kernel<3>), resident_blocks(wgrad_thin_kernel<4>)}};
  return c;
}

int conv3x3_thin_ok(int cin, int cout_p) {
  if (!(thin_enabled() && cin >= 1 && cin <= 4 && cout_p % 4 == 0 && cout_p >= 4 && cout_p <= 256)) return 0;
  (void)thin_caps();
  return 1;
}

int conv3x3_thin_launch(const ConvLaunch& a, int cin, int* rows, hipStream_t stream) {
  if (!conv3x3_thin_ok(cin, a.cout_store) || a.off != 1 || a.Hi != a.Ho || a.Wi != a.Wo || a.Hi < 2 || a.Wi < 2 || a.cin_p < cin ||
      a.ldx % 4 != 0 || a.ldy % 4 != 0 || (int64_t)a.N * a.Hi * a.Wi > INT32_MAX) {
    set_error("conv3x3_thin: bad geometry cin=%d cout=%d H=%d W=%d", cin, a.cout_store, a.Hi, a.Wi);
    return MIMO_ERR_INVALID;
  }
  const int Cv = a.cout_store / 4;
  const int tilesY = (a.Hi + kTH - 1) / kTH, tilesX = (a.Wi + kTW - 1) / kTW, numTiles = a.N * tilesY * tilesX;
  const int blocks = std::min(numTiles, thin_caps().fwd[cin - 1]);
  if (rows) *rows = blocks;
  switch (cin) {
    case 1: hipLaunchKernelGGL(conv3x3_thin_fwd_kernel<1>, dim3(blocks), dim3(256), 0, stream, a, Cv, tilesY, tilesX, numTiles); break;
    case 2: hipLaunchKernelGGL(conv3x3_thin_fwd_kernel<2>, dim3(blocks), dim3(256), 0, stream, a, Cv, tilesY, tilesX, numTiles); break;
    case 3: hipLaunchKernelGGL(conv3x3_thin_fwd_kernel<3>, dim3(blocks), dim3(256), 0, stream, a, Cv, tilesY, tilesX, numTiles); break;
    default: hipLaunchKernelGGL(conv3x3_thin_fwd_kernel<4>, dim3(blocks), dim3(256), 0, stream, a, Cv, tilesY, tilesX, numTiles); break;
  }
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

size_t wgrad_thin_scratch(int cin, int cout_p) { return (size_t)kThinBlocks * 9 * cin * cout_p; }

// The weight gradient pays on the plain-FMA kernel only with <= 2 input channels (36 accumulator registers per channel) and
// enough tiles to pipeline (measured: 2 -> 30 at 256x256, batch 32: 94 us against 125 us on the split kernel; 3 -> 21: 136
// against 123; batch 4: 26 against 22)
int wgrad_thin_ok(int cin, int cout_p, int N, int H, int W) {
  if (!conv3x3_thin_ok(cin, cout_p) || cin > 2) return 0;
  const int64_t tiles = (int64_t)N * ((H + kTH - 1) / kTH) * ((W + kTW - 1) / kTW);
  return (tiles >= 4 * (int64_t)thin_caps().wg[cin - 1] || thin_level() == 2) ? 1 : 0;
}

int wgrad_thin_launch(const float* x, int ldx, const float* dz, int lddz, int N, int H, int W, int cin, int cout, int cout_p,
                      float* partial, float* dw, hipStream_t stream) {
  if (!conv3x3_thin_ok(cin, cout_p) || H < 2 || W < 2 || ldx % 4 != 0 || lddz % 4 != 0 || (int64_t)N * H * W > INT32_MAX) {
    set_error("wgrad_thin: bad geometry cin=%d cout_p=%d H=%d W=%d", cin, cout_p, H, W);
    return MIMO_ERR_INVALID;
  }
  const int Cv = cout_p / 4;
  const int tilesY = (H + kTH - 1) / kTH, tilesX = (W + kTW - 1) / kTW, numTiles = N * tilesY * tilesX;
  const int blocks = std::min(numTiles, thin_caps().wg[cin - 1]);
  switch (cin) {
    case 1: hipLaunchKernelGGL(wgrad_thin_kernel<1>, dim3(blocks), dim3(256), 0, stream, x, ldx, dz, lddz, N, H, W, Cv, tilesY, tilesX, numTiles, partial); break;
    case 2: hipLaunchKernelGGL(wgrad_thin_kernel<2>, dim3(blocks), dim3(256), 0, stream, x, ldx, dz, lddz, N, H, W, Cv, tilesY, tilesX, numTiles, partial); break;
    case 3: hipLaunchKernelGGL(wgrad_thin_kernel<3>, dim3(blocks), dim3(256), 0, stream, x, ldx, dz, lddz, N, H, W, Cv, tilesY, tilesX, numTiles, partial); break;
    default: hipLaunchKernelGGL(wgrad_thin_kernel<4>, dim3(blocks), dim3(256), 0, stream, x, ldx, dz, lddz, N, H, W, Cv, tilesY, tilesX, numTiles, partial); break;
  }
  MIMO_KERNEL_CHECK();
  hipLaunchKernelGGL(wgrad_thin_reduce_kernel, dim3(9 * cin), dim3(256), 0, stream, partial, blocks, cin, cout_p, cout, dw);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

}  // namespace mimo
