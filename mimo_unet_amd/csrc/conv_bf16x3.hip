// 3x3 convolution (forward / data gradient) on the 16-bit MFMA with split operands:
//   a = a_hi + a_lo + eps,  a.b ~= a_hi.b_hi + a_hi.b_lo + a_lo.b_hi   (fp32 accumulate)
// Three v_mfma_f32_16x16x32_{f16,bf16} per product block = 16/3 = 5.3x the rate of the f32-input
// MFMA that bounds conv3x3.hip.  Two element types, chosen by what the operand tolerates:
//   * F16 = true  (forward): fp16 hi/lo carry 11+11 mantissa bits -> ~2^-22 per product, i.e.
//     fp32-class outputs.  That matters: a 5e-5 forward error (bf16 hi/lo) flips enough ReLU /
//     max-pool masks to move small-network gradients by tens of percent (measured on CPU by
//     emulation).  Weights are pre-scaled by 2^8 (kept out of the fp16 subnormal range; undone
//     exactly in the epilogue); activations below 0.125 keep >= 3e-8 absolute precision.
//   * F16 = false (data gradient): bf16 hi/lo keep fp32's exponent range for the tiny dz values;
//     ~1e-5 per product on the gradient side is harmless (gradient error 2e-5, same emulation).
//     The input (dz) arrives PRE-SPLIT — per pixel and 32-channel chunk [hi 32 | lo 32] bf16, written
//     by the BatchNorm-backward kernel (elementwise.hip st_split4) — which is this kernel's LDS row
//     image: the loader is a plain copy of one 128-byte line per (pixel, chunk).
//
// Same implicit GEMM as conv3x3.hip (M = TRxTC output pixels of one image, N = output channels,
// K = 9 taps x input channels) with a deeper structure:
//   * workgroup = 512 threads = 8 waves, one per-SIMD pair; tile = 8*MF fragments of 16 pixels
//     (256 or 512 pixels) x NF*16 output channels; MF x NF accumulator fragments per wave
//   * K is walked in 32-channel chunks x 3 tap rows ("phases").  The input halo tile of a chunk is
//     staged once (fp32 -> hi/lo bf16 split on the way into LDS); the weights of one tap row
//     (3 taps x NB x 32 ch, pre-split by the packer) are double-buffered in LDS
//   * every phase first ISSUES the global loads of the next phase's weights (and, in the first
//     phase of a chunk, of the next chunk's input tile) into registers, then runs its MFMAs, then
//     writes the prefetched registers to LDS: global latency hides under the MFMA work
//   * LDS rows are 64 bf16 (32 hi | 32 lo) + 16 B pad = 144 B: ds_read_b128 fragment reads with
//     compile-time offsets, 16 consecutive pixels land on distinct bank quads
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace mimo {

template <bool F16>
struct Elem;
template <>
struct Elem<false> {
  typedef __bf16 T;
  typedef __attribute__((ext_vector_type(8))) __bf16 V8;
  typedef __attribute__((ext_vector_type(4))) __bf16 V4;
};
template <>
struct Elem<true> {
  typedef _Float16 T;
  typedef __attribute__((ext_vector_type(8))) _Float16 V8;
  typedef __attribute__((ext_vector_type(4))) _Float16 V4;
};
typedef float f32x4_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_ mfma16(Elem<false>::V8 a, Elem<false>::V8 b, f32x4_ c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4_ mfma16(Elem<true>::V8 a, Elem<true>::V8 b, f32x4_ c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
constexpr float kF16WeightScale = 256.f;  // 2^8, exact

#ifdef MIMO_CONV_STAMPS
// timing-only instrumentation (-DMIMO_CONV_STAMPS): per kernel role, shader-clock ticks summed over one wave of every
// workgroup — [0] forward consumers: barrier wait, [1] their total, [2] forward producers: wait (DMA / barrier), [3]
// their total, [4..7] the same for the data gradient
__device__ unsigned long long g_conv_stamps[8];
#define STAMP_NOW() __builtin_amdgcn_s_memtime()
#endif

// Kernel modes (template parameter MODE of the convolution kernels and launchers):
//   0  split16 data gradient : bf16 (hi, lo) pairs, input pre-split (dz),       3 MFMAs per product
//   1  split16 forward       : fp16 (hi, lo) pairs, fp32 input split on the way, 3 MFMAs per product
//   2  bf16 forward          : bf16 hi only,        fp32 input rounded on the way, 1 MFMA per product
//   3  bf16 data gradient    : bf16 hi only,        input pre-split (lo ignored),  1 MFMA per product
//   4  bf16-mixed forward    : bf16, input AND output stored as plain NHWC bf16,  1 MFMA per product
//   5  bf16-mixed data grad  : same, dz / dx stored as bf16
//   6  fp16-mixed forward    : fp16 storage and operands (the reference's precision="16-mixed"), weights x 2^8
//   7  fp16-mixed data grad  : same (gradients arrive multiplied by the loss scale)
// (modes 2 / 3 = mimo_precision BF16: "bf16 compute, fp32 accumulate"; modes >= 2 keep the [hi 32 | lo 32] LDS row
// image and simply leave the lo half unused)
#define MIMO_CONV_MODE_CONSTANTS                                                                         \
  constexpr bool F16 = MODE == 1 || MODE >= 6;  /* element type fp16 (else bf16) */                       \
  constexpr bool CVT = MODE == 1 || MODE == 2;  /* loader converts fp32 input */                          \
  constexpr bool IN16 = MODE >= 4;              /* input = plain NHWC of the 16-bit element type */       \
  constexpr bool OUT16 = MODE >= 4;             /* output stored in the 16-bit element type */            \
  constexpr bool FWD = MODE == 1 || MODE == 2 || MODE == 4 || MODE == 6; /* bias, BatchNorm sums, inference epilogue */ \
  constexpr int NP = MODE >= 2 ? 1 : 3;         /* MFMAs per product block */
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kPitchB = 144;  // bytes per LDS row (pixel or weight row)

using sched::pick_tile_n;  // tile_sched.h (host-testable)

template <int MF>
struct TileCfg {
  static constexpr int NPIX = 8 * MF * 16;
  static constexpr int MAXPIX = MF == 4 ? 640 : 360;
};

template <int MF, int NF, int MODE>
__global__ __launch_bounds__(512) void conv3x3_bf16x3_kernel(ConvLaunch a, int TR, int TC, int tilesY, int tilesX) {
  MIMO_CONV_MODE_CONSTANTS
  (void)FWD;  // this kernel's epilogue keys on the argument pointers (bias / stats / ep_scale null for the data gradient)
  typedef typename Elem<F16>::T ET;
  typedef typename Elem<F16>::V8 bf16x8;
  typedef typename Elem<F16>::V4 bf16x4;
  constexpr int NB = NF * 16;
  constexpr int MAXPIX = TileCfg<MF>::MAXPIX;
  constexpr int XU = (MAXPIX * 8 + 511) / 512;      // float4 units of the input tile per thread
  constexpr int WUNITS = 3 * NB * 8;                 // 16-byte units of one weight tap row
  constexpr int WU = (WUNITS + 511) / 512;
  constexpr int WROWB = 3 * NB * kPitchB;            // bytes of one weight row buffer
  __shared__ __attribute__((aligned(16))) unsigned char xs[MAXPIX * kPitchB];
  __shared__ __attribute__((aligned(16))) unsigned char ws[2 * WROWB];
  __shared__ int goff[MAXPIX];

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

  for (int p = tid; p < npix_lds; p += 512) {
    const int tr = p / TCP, tc = p - tr * TCP;
    int iy = y0 - a.off + tr, ix = x0 - a.off + tc;
    int o;
    if (a.off == 1) {
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
  __syncthreads();

  const float* ximg = a.x + (size_t)n * a.Hi * a.Wi * a.ldx;
  const unsigned short* ximg16 = reinterpret_cast<const unsigned short*>(a.x) + (size_t)n * a.Hi * a.Wi * a.ldx;  // IN16
  const int nchunks = (a.cin_p + 31) / 32;
  const u32x4* wpk = reinterpret_cast<const u32x4*>(a.wpk);

  // ---- staging helpers (plain unrolled code: the prefetch arrays must stay in registers) ------
  f32x4 xreg[XU];
  u32x4 wreg[3][WU];  // weights in flight for the next TWO phases (memory latency under load > one phase)
#define MIMO_LOAD_X(CHUNK)                                                                      \
  _Pragma("unroll") for (int k_ = 0; k_ < XU; ++k_) {                                           \
    const int u_ = tid + k_ * 512;                                                              \
    const int p_ = min(u_ >> 3, npix_lds - 1), q_ = u_ & 7;                                     \
    const int o_ = goff[p_];                                                                    \
    /* masked-out units are loaded from kZeroPage (common.h), never selected after the load */   \
    if (CVT) { /* fp32 input: unit q_ = 4 channels */                                           \
      const int ch_ = (CHUNK) * 32 + 4 * q_;                                                    \
      const bool ok_ = o_ >= 0 && ch_ < a.cin_p;                                                \
      xreg[k_] = *reinterpret_cast<const f32x4*>(ok_ ? ximg + o_ + ch_ : kZeroPage);            \
    } else if (IN16) { /* plain 16-bit NHWC: units 0..3 = 8 channels each (the hi half of the row), 4..7 unused */ \
      const int rc_ = min(32, a.cin_p - (CHUNK) * 32);                                          \
      const bool ok_ = o_ >= 0 && q_ < 4 && 8 * q_ < rc_;                                       \
      xreg[k_] = *reinterpret_cast<const f32x4*>(                                               \
          ok_ ? reinterpret_cast<const float*>(ximg16 + o_ + (CHUNK) * 32 + 8 * q_) : kZeroPage); \
    } else { /* pre-split input, chunk record [hi rc | lo rc] bf16: unit q_ = 8 channels, 16 bytes */ \
      const int rc_ = min(32, a.cin_p - (CHUNK) * 32);                                          \
      const bool ok_ = o_ >= 0 && 8 * (q_ & 3) < rc_;                                           \
      const unsigned short* s_ = reinterpret_cast<const unsigned short*>(ximg + o_) +           \
                                 ((CHUNK) * 64 + (q_ >> 2) * rc_ + 8 * (q_ & 3));               \
      xreg[k_] = *reinterpret_cast<const f32x4*>(ok_ ? reinterpret_cast<const float*>(s_) : kZeroPage); \
    }                                                                                           \
  }
#define MIMO_STORE_X()                                                                          \
  _Pragma("unroll") for (int k_ = 0; k_ < XU; ++k_) {                                           \
    const int u_ = tid + k_ * 512;                                                              \
    const int p_ = u_ >> 3, q_ = u_ & 7;                                                        \
    if (p_ < npix_lds) {                                                                        \
      const f32x4 v_ = xreg[k_];                                                                \
      if (CVT) {                                                                                \
        bf16x4 hi_, lo_;                                                                        \
        hi_[0] = (ET)v_[0];                                                                     \
        hi_[1] = (ET)v_[1];                                                                     \
        hi_[2] = (ET)v_[2];                                                                     \
        hi_[3] = (ET)v_[3];                                                                     \
        lo_[0] = (ET)(v_[0] - (float)hi_[0]);                                                   \
        lo_[1] = (ET)(v_[1] - (float)hi_[1]);                                                   \
        lo_[2] = (ET)(v_[2] - (float)hi_[2]);                                                   \
        lo_[3] = (ET)(v_[3] - (float)hi_[3]);                                                   \
        unsigned char* d_ = xs + p_ * kPitchB + q_ * 8;                                         \
        *reinterpret_cast<bf16x4*>(d_) = hi_;                                                   \
        if (NP == 3) *reinterpret_cast<bf16x4*>(d_ + 64) = lo_;                                 \
      } else { /* the row image [hi 32 | lo 32] is the global layout: plain 16-byte copy */     \
        *reinterpret_cast<f32x4*>(xs + p_ * kPitchB + q_ * 16) = v_;                            \
      }                                                                                         \
    }                                                                                           \
  }
  // weights of chunk CHUNK, tap row ROW: global [chunk][tap][cout_pad][8 x 16 B]
#define MIMO_LOAD_W(SLOT, CHUNK, ROW)                                                           \
  _Pragma("unroll") for (int k_ = 0; k_ < WU; ++k_) {                                           \
    const int u_ = min(tid + k_ * 512, WUNITS - 1);                                             \
    const int t_ = u_ / (NB * 8);                                                               \
    const int rem_ = u_ - t_ * (NB * 8);                                                        \
    wreg[SLOT][k_] = wpk[(((size_t)(CHUNK) * 9 + ((ROW) * 3 + t_)) * a.cout_pad + co0 + (rem_ >> 3)) * 8 + (rem_ & 7)]; \
  }
#define MIMO_STORE_W(SLOT, BUF)                                                                 \
  _Pragma("unroll") for (int k_ = 0; k_ < WU; ++k_) {                                           \
    const int u_ = tid + k_ * 512;                                                              \
    if (u_ < WUNITS) {                                                                          \
      const int t_ = u_ / (NB * 8);                                                             \
      const int rem_ = u_ - t_ * (NB * 8);                                                      \
      *reinterpret_cast<u32x4*>(ws + (BUF) * WROWB + (t_ * NB + (rem_ >> 3)) * kPitchB + (rem_ & 7) * 16) = wreg[SLOT][k_]; \
    }                                                                                           \
  }

  // ---- per-lane fragment bases -----------------------------------------------------------
  int pbase[MF];
#pragma unroll
  for (int m = 0; m < MF; ++m) {
    int idx = (wave * MF + m) * 16 + lr;
    if (idx >= npix_out) idx = 0;
    const int r = idx / TC, c = idx - r * TC;
    pbase[m] = (r * TCP + c) * kPitchB + g * 16;
  }
  const int wbase = lr * kPitchB + g * 16;

  f32x4 acc[MF][NF];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[m][nf] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: chunk 0, tap row 0
  // phase p = 3*c + r uses weight slot r; slot (r+1)%3 holds phase p+1 (in registers), slot (r+2)%3
  // is being loaded for phase p+2
  const int nphases = 3 * nchunks;
  MIMO_LOAD_X(0)
  MIMO_LOAD_W(0, 0, 0)
  if (nphases > 1) {
    MIMO_LOAD_W(1, 0, 1)
  }
  MIMO_STORE_X()
  MIMO_STORE_W(0, 0)
  __syncthreads();

  int buf = 0;
  for (int c = 0; c < nchunks; ++c) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int ph = 3 * c + r;
      const bool last = ph + 1 >= nphases;
      if (ph + 2 < nphases) {  // two phases ahead: (c, r+2) or (c+1, r-1)
        const int nc = r < 1 ? c : c + 1, nr = (r + 2) % 3;
        MIMO_LOAD_W((r + 2) % 3, nc, nr)
      }
      if (r == 0 && c + 1 < nchunks) {
        MIMO_LOAD_X(c + 1)
      }
      const unsigned char* wb = ws + buf * WROWB + wbase;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int toff = (r * TCP + kw) * kPitchB;
        bf16x8 bh[NF], bl[NF];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          bh[nf] = *reinterpret_cast<const bf16x8*>(wb + (kw * NB + nf * 16) * kPitchB);
          if (NP == 3) bl[nf] = *reinterpret_cast<const bf16x8*>(wb + (kw * NB + nf * 16) * kPitchB + 64);
        }
#pragma unroll
        for (int m = 0; m < MF; ++m) {
          const bf16x8 ah = *reinterpret_cast<const bf16x8*>(xs + pbase[m] + toff);
          bf16x8 al = ah;
          if (NP == 3) al = *reinterpret_cast<const bf16x8*>(xs + pbase[m] + toff + 64);
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) {
            if (NP == 3) {
              acc[m][nf] = mfma16(al, bh[nf], acc[m][nf]);
              acc[m][nf] = mfma16(ah, bl[nf], acc[m][nf]);
            }
            acc[m][nf] = mfma16(ah, bh[nf], acc[m][nf]);
          }
        }
      }
      if (!last) {  // other buffer: last read one phase ago, behind a barrier
        MIMO_STORE_W((r + 1) % 3, buf ^ 1)
      }
      if (r == 2 && c + 1 < nchunks) {
        __syncthreads();  // every wave is done reading this chunk's input tile
        MIMO_STORE_X()
      }
      __syncthreads();
      buf ^= 1;
    }
  }

  // ---- epilogue: bias, store, BatchNorm partial sums -----------------------------------------
  float bv[NF], s1[NF], s2[NF], esc[NF], esh[NF], emk[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    bv[nf] = a.bias ? a.bias[co0 + nf * 16 + lr] : 0.f;
    s1[nf] = 0.f;
    s2[nf] = 0.f;
    esc[nf] = a.ep_scale ? a.ep_scale[co0 + nf * 16 + lr] : 1.f;
    esh[nf] = a.ep_scale ? a.ep_shift[co0 + nf * 16 + lr] : 0.f;
    emk[nf] = (a.ep_mask && co0 + nf * 16 + lr < a.ep_mask_ld) ? a.ep_mask[(size_t)n * a.ep_mask_ld + co0 + nf * 16 + lr] : 1.f;
  }
  typedef typename std::conditional<OUT16, ET, float>::type OT;
  OT* yimg = reinterpret_cast<OT*>(a.y) + (size_t)n * a.Ho * a.Wo * a.ldy;
  const float winv = !F16 ? 1.f : a.wmax ? w16_scale(*a.wmax, true) : 1.f / kF16WeightScale;  // the fp16 image's scale
#pragma unroll
  for (int m = 0; m < MF; ++m) {
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int idx = (wave * MF + m) * 16 + g * 4 + r4;
      const int orow = idx / TC, ocol = idx - orow * TC;
      const int oy = y0 + orow, ox = x0 + ocol;
      if (idx < npix_out && oy < a.Ho && ox < a.Wo) {
        OT* yp = yimg + ((size_t)oy * a.Wo + ox) * a.ldy + co0 + lr;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          float v = acc[m][nf][r4] * winv + bv[nf];
          if (a.ep_scale) {
            if (a.status && !isfinite(v)) atomicOr(a.status, 1);  // the ReLU below would drop a NaN
            v = fmaxf(fmaf(v, esc[nf], esh[nf]), 0.f) * emk[nf];
          }
          if (co0 + nf * 16 + lr < a.cout_store) yp[nf * 16] = (OT)v;
          s1[nf] += v;
          s2[nf] += v * v;
        }
      }
    }
  }
  if (a.stats) {
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      s1[nf] += __shfl_xor(s1[nf], 16);
      s1[nf] += __shfl_xor(s1[nf], 32);
      s2[nf] += __shfl_xor(s2[nf], 16);
      s2[nf] += __shfl_xor(s2[nf], 32);
    }
    float* red = reinterpret_cast<float*>(ws);  // [8 waves][2][NB]; the main loop ended on a barrier
    if (g == 0) {
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        red[(wave * 2 + 0) * NB + nf * 16 + lr] = s1[nf];
        red[(wave * 2 + 1) * NB + nf * 16 + lr] = s2[nf];
      }
    }
    __syncthreads();
    if (tid < 2 * NB) {
      const int which = tid / NB, c = tid - which * NB;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += red[(w * 2 + which) * NB + c];
      a.stats[((size_t)blockIdx.x * 2 + which) * a.cout_pad + co0 + c] = v;
    }
  }
}

#undef MIMO_LOAD_X
#undef MIMO_STORE_X
#undef MIMO_LOAD_W
#undef MIMO_STORE_W

// ---------------------------------------------------------------------------------------
// Wave-specialised, persistent variant (same arithmetic, same packed weights, same LDS row images).
// In the kernel above every wave stages AND multiplies, and the input tile of the next chunk is
// rewritten in an exclusive barrier-to-barrier section.  Here waves 0-3 ("consumers") only issue
// ds_read_b128 + MFMA and waves 4-7 ("producers", one next to a consumer on every SIMD) do all the
// staging: global loads two stages ahead in registers, fp32 -> hi/lo split (forward) or plain copy
// (data gradient, pre-split dz), LDS writes into the OTHER input-tile / weight buffer.  The
// workgroup is persistent over output tiles, so the first chunk of the next tile is staged under the
// last chunk of the current one and the output epilogue of the consumers overlaps the producers.
//   tile = 256 output pixels (4 consumers x 4 fragments of 16 pixels) x NF*16 output channels
//   LDS  = 2 input tiles (360 x 144 B) + 2 weight tap rows (3 x NB x 144 B) <= 159 KB
//   one workgroup barrier per phase (tap row of a 32-channel chunk)
// BatchNorm partial sums: accumulated in registers over the workgroup's tiles, the four consumer waves combined
// through LDS once at the end: one row per workgroup for the column reduction that follows.
// ---------------------------------------------------------------------------------------
// MF = fragments of 16 pixels per consumer wave.  4: 256-pixel tiles, one workgroup per CU.  2: 128-pixel tiles
// (180-pixel halo); with NF <= 2 the workgroup needs 79 KB of LDS and 128 registers per lane, so TWO workgroups
// share a CU and one's MFMAs fill the other's barrier / staging / epilogue bubbles — for the thin layers (<= 32
// output channels), whose tiles carry too little MFMA work to hide those.
template <int MF>
struct WsTile {
  static constexpr int NPIX = 64 * MF;
  static constexpr int MAXPIX = MF == 4 ? 360 : 180;
};
constexpr int kWsMaxMF = 4;

// PAIR (split16 modes, padded rows): the LAST 32-channel chunk of the input holds <= 16 channels.  Its taps are then
// multiplied two per MFMA in TWO phases of three steps instead of three: step kw of the first phase carries channels
// 0-15 of tap (0, kw) in K lanes 0-15 and of tap (1, kw) in K lanes 16-31, step kw of the second tap (2, kw) in K
// lanes 0-15 (weights of lanes 16-31 zero); the chunk's third phase is a bare barrier (the producers stage as
// before).  45 input channels: 15 K steps per tile instead of 18.  The weight image of the chunk is packed accordingly
// (conv3x3_pair_tail, pack kernels below); the input tile in LDS is unchanged: the lanes of K groups 2-3 read the
// pixel one tile row further down at slots 0-1 — a per-lane constant, so the fragment addresses keep the form
// (lane base) + (scalar of the phase) + (immediate of the step), and phases keep their three steps, so the
// alternation of the two B register sets is the same compile-time pattern as without pairing.
//
// WDMA: the weights of a phase go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write) into
// the buffer the consumers released at the last barrier, one phase ahead; the producers wait for them with a counted
// vmcnt that leaves their own (younger) input-tile loads in flight, then pass a raw s_barrier.  A DMA instruction
// writes 64 x 16 contiguous bytes, so the weight rows are unpadded 128-byte rows with the slot index XORed by
// (row & 7) (conflict-free ds_read_b128, as SWZ); the XOR is applied to the per-lane SOURCE address.
//
// FIN (split16 forward, grad-enabled path; round 4): the input is the producing convolution's pre-activation tensor and the
// producers apply its BatchNorm + ReLU (ConvLaunch::in_scale / in_shift: relu(fma(z, scale, shift)), bn_relu_fwd_kernel's
// arithmetic) in front of the fp16 split, so the activated tensor between the two convolutions of a DoubleConv
// (components.py:24-25) is never written.  A thread's units all hold one channel quad of a chunk; its constants are loaded
// one stage ahead, in front of that phase's weight DMA and input loads (the counted vmcnt waits stay valid).
template <int NF, int MODE, int MF_, bool PAIR = false, bool WDMA = true, bool FIN = false>
__global__ __launch_bounds__(512, (MF_ == 2 && NF <= 2) ? 4 : 2) void conv3x3_ws_kernel(ConvLaunch a, int TR, int TC,
                                                                                          int tilesY, int tilesX,
                                                                                          int numTiles, int xcd_order,
                                                                                          int gx, int coTiles) {
  MIMO_CONV_MODE_CONSTANTS
  static_assert(!PAIR || (NP == 3 && !IN16), "tap pairing: split16 modes");
  static_assert(!FIN || (MODE == 1 && WDMA), "fused input BatchNorm + ReLU: the split16 forward, weights by DMA");
  typedef typename Elem<F16>::T ET;
  typedef typename Elem<F16>::V8 bf16x8;
  typedef typename Elem<F16>::V4 bf16x4;
  typedef typename std::conditional<OUT16, ET, float>::type OT;  // element type of the output tensor
  constexpr int NB = NF * 16;
  constexpr int MF = MF_;
  constexpr int kWsMaxPix = WsTile<MF>::MAXPIX;
  // 16-byte units per LDS row that carry data: 8 (fp32 input: 4 channels each; pre-split input: hi + lo halves) or,
  // for plain 16-bit input, the 4 units of the hi half (8 channels each)
  constexpr int UPP = IN16 ? 4 : 8, UPPS = IN16 ? 2 : 3;
  constexpr int XU = (kWsMaxPix * UPP + 255) / 256;  // units of an input tile per producer thread (12 / 6; IN16: 6 / 3)
  constexpr int XP = XU / 3;                          // units handled per phase
  static_assert(XU % 3 == 0, "input tile staged in three equal parts");
  constexpr int WUNITS = 3 * NB * 8;
  constexpr int WU = (WUNITS + 255) / 256;
  // LDS row = [hi 32 | lo 32] 16-bit values = eight 16-byte slots in a 144-byte padded row (2-way conflicting on the
  // ds_read_b128 of 16 consecutive rows; conflict-free XOR-swizzled 128-byte input rows measured the same kernel
  // times in rounds 1 and 2 and were removed in round 3)
  constexpr int PITCH = kPitchB;
  constexpr bool WSWZ = WDMA;                      // weight rows: swizzled 128-byte rows (the DMA writes 1 KB runs)
  constexpr int WPITCH = WSWZ ? 128 : kPitchB;
  constexpr int XBYTES = kWsMaxPix * PITCH, WROWB = 3 * NB * WPITCH;
  __shared__ __attribute__((aligned(128))) unsigned char xs[2 * XBYTES];
  __shared__ __attribute__((aligned(128))) unsigned char ws[2 * WROWB];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int TCP = TC + 2, TRP = TR + 2;
  const int npix_lds = TRP * TCP, npix_out = TR * TC;
  const int nchunks = (a.cin_p + 31) / 32;
  // 1-D grid of gx * coTiles workgroups: gx persistent pixel-tile columns x coTiles output-channel tiles.
  // XCD-aware order (xcd_virtual_index, common.h): virtual index v = column * coTiles + channel tile, so the
  // coTiles workgroups that read the SAME input tiles sit on one XCD and fetch them from HBM once (measured
  // before: 11 x re-read of the input on the 480-channel layers, one per channel tile), and an XCD owns
  // gx / 8 CONSECUTIVE columns — vertically adjacent tile rows, whose halo rows they share.
#ifdef MIMO_CONV_ABLATE
  // timing-only builds (-DMIMO_CONV_ABLATE=bits, results are wrong; WDMA instances only): 1 = producers skip the
  // input-tile loads, 2 = the input-tile LDS stores, 4 = the weight staging; 8 = consumers skip the MFMAs, 16 = the
  // fragment reads, 32 = the per-tile epilogue (stores, bias / statistics arithmetic; the accumulators stay live)
  constexpr int abl = MIMO_CONV_ABLATE;
#else
  constexpr int abl = 0;
#endif
  // K split (ConvLaunch::ksplit > 1; round 6): the workgroups of a (pixel-tile column, channel tile) pair come ksplit-fold, each
  // walks nck consecutive 32-channel chunks from ck0 and stores its partial sums into slab ks of a.y; ksplit == 1: ck0 = 0,
  // nck = nchunks, one slab — the arithmetic below is then the old one.  Virtual order: (column, split, channel tile), so
  // the channel tiles that read the same input chunks stay neighbours.
  const int ksplit = a.ksplit > 1 ? a.ksplit : 1;
  const int v_ = xcd_order ? xcd_virtual_index((int)blockIdx.x, gx * coTiles * ksplit) : (int)blockIdx.x;
  const int vks_ = v_ / coTiles;
  const int vbx = vks_ / ksplit, ks = vks_ - vbx * ksplit;
  const int co0 = (v_ - vks_ * coTiles) * NB;
  const int cpk_ = (nchunks + ksplit - 1) / ksplit;
  const int ck0 = ks * cpk_, nck = min(cpk_, nchunks - ck0);  // (the launch guarantees nck >= 1)
  const int ntiles_mine = vbx < numTiles ? (numTiles - 1 - vbx) / gx + 1 : 0;
  const int nstages = ntiles_mine * nck;  // input-tile stages (tile, chunk); 3 phases each

  if (wave >= 4) {
    // =============================== producers ===============================
    const int ptid = tid - 256;
    f32x4 xreg[XU];
    u32x4 wreg[WDMA ? 1 : 3][WDMA ? 1 : WU];
    // WDMA: 16-byte unit (u & 7) ^ (row & 7) of weight row u >> 3 of the phase lands at LDS unit u = ptid + 256 k
    const unsigned ws_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ws;  // LDS byte address
    int wsrc[WU];
#pragma unroll
    for (int k = 0; k < WU; ++k) {
      const int u = min(ptid + k * 256, WUNITS - 1), row = u >> 3;
      wsrc[k] = ((row / NB) * a.cout_pad + row % NB) * 8 + ((u & 7) ^ (row & 7));
    }
    // halo-tile coordinates of this thread's units (tile-shape constants): (row << 16) | column, and the byte
    // offset of the unit from the tile's first halo pixel for tiles whose halo lies inside the image
    int u_rc[XU], u_off[XU];
#pragma unroll
    for (int k = 0; k < XU; ++k) {
      const int p = min((ptid + k * 256) >> UPPS, npix_lds - 1);
      const int tr = p / TCP, tc = p - tr * TCP;
      u_rc[k] = (tr << 16) | tc;
      u_off[k] = ((tr * a.Wi + tc) * a.ldx + 4 * ((ptid + k * 256) & 7)) * 4;  // pre-split input only
    }
    const u32x4* wpk = reinterpret_cast<const u32x4*>(a.wpk);
    // FIN: BatchNorm scale / shift of the thread's channel quad, of the stage being stored (in_sc / in_sh) and of the stage
    // being loaded (in_sc_n / in_sh_n); channels past cin_p come from the zero page: relu(0 * 0 + 0) keeps them zero
    f32x4 in_sc = f32x4{0.f, 0.f, 0.f, 0.f}, in_sh = in_sc, in_sc_n = in_sc, in_sh_n = in_sc;
#define WS_LOAD_SS(SC, SH, STAGE)                                                                    \
  if (FIN) {                                                                                         \
    const int c_ = (ck0 + (STAGE) % nck) * 32 + 4 * (ptid & 7);                                      \
    /* UNCONDITIONAL (kFinSlack zero floats behind the layer's channels, plan.hip): exactly two loads per stage */ \
    SC = *reinterpret_cast<const f32x4*>(a.in_scale + c_);                                           \
    SH = *reinterpret_cast<const f32x4*>(a.in_shift + c_);                                           \
  }
    // stage j = (tile j / nchunks, chunk j % nchunks)
#define WS_LOAD_X(K0, K1, STAGE)                                                                     \
  {                                                                                                  \
    const int st_ = (STAGE);                                                                         \
    const int ti_ = st_ / nck, ck_ = ck0 + st_ - ti_ * nck;                                          \
    int t_ = vbx + ti_ * gx;                                                                         \
    const int tx_ = t_ % tilesX;                                                                     \
    t_ /= tilesX;                                                                                    \
    const int ty_ = t_ % tilesY;                                                                     \
    const int n_ = t_ / tilesY;                                                                      \
    const int y0_ = ty_ * TR - a.off, x0_ = tx_ * TC - a.off;                                        \
    const float* ximg_ = a.x + (size_t)n_ * a.Hi * a.Wi * a.ldx;                                     \
    const unsigned short* ximg16_ = reinterpret_cast<const unsigned short*>(a.x) + (size_t)n_ * a.Hi * a.Wi * a.ldx; \
    /* halo inside the image and a full 32-channel chunk (a wave-uniform test): no reflection, no     \
       masking, address = scalar tile base + per-thread constant -- the ~20 vector instructions per    \
       unit of the general path compete with the MFMAs for the SIMD's issue port.  Data gradient     \
       only: measured -8 % on its thin layers, nothing on the forward (whose producers are bound by   \
       the fp32 -> fp16 split) */                                                                     \
    if (!CVT && !IN16 && y0_ >= 0 && y0_ + TRP <= a.Hi && x0_ >= 0 && x0_ + TCP <= a.Wi && ck_ * 32 + 32 <= a.cin_p) { \
      const char* tb_ = reinterpret_cast<const char*>(ximg_ + ((size_t)y0_ * a.Wi + x0_) * a.ldx) + ck_ * 128; \
      _Pragma("unroll") for (int k_ = (K0); k_ < (K1); ++k_)                                         \
        xreg[k_] = *reinterpret_cast<const f32x4*>(tb_ + u_off[k_]);                                 \
    } else {                                                                                         \
    _Pragma("unroll") for (int k_ = (K0); k_ < (K1); ++k_) {                                         \
      int iy_ = y0_ + (u_rc[k_] >> 16), ix_ = x0_ + (u_rc[k_] & 0xffff);                             \
      bool in_ = true;                                                                               \
      if (a.off == 1) { /* reflect (forward) */                                                      \
        iy_ = max(iy_, -iy_);                                                                        \
        iy_ = max(min(iy_, 2 * a.Hi - 2 - iy_), 0);                                                  \
        ix_ = max(ix_, -ix_);                                                                        \
        ix_ = max(min(ix_, 2 * a.Wi - 2 - ix_), 0);                                                  \
      } else { /* zero padding (transposed convolution) */                                           \
        in_ = iy_ >= 0 && iy_ < a.Hi && ix_ >= 0 && ix_ < a.Wi;                                      \
      }                                                                                              \
      const int q_ = (ptid + k_ * 256) & (UPP - 1);                                                  \
      const int o_ = in_ ? (iy_ * a.Wi + ix_) * a.ldx : 0;                                           \
      if (IN16) { /* plain 16-bit NHWC: unit q_ = 8 channels of the chunk */                         \
        const int rc_ = min(32, a.cin_p - ck_ * 32);                                                 \
        const bool ok_ = in_ && 8 * q_ < rc_;                                                        \
        xreg[k_] = *reinterpret_cast<const f32x4*>(                                                  \
            ok_ ? reinterpret_cast<const float*>(ximg16_ + o_ + ck_ * 32 + 8 * q_) : kZeroPage);     \
      } else if (CVT) {                                                                              \
        const int ch_ = ck_ * 32 + 4 * q_;                                                           \
        const bool ok_ = in_ && ch_ < a.cin_p;                                                       \
        xreg[k_] = *reinterpret_cast<const f32x4*>(ok_ ? ximg_ + o_ + ch_ : kZeroPage);              \
      } else {                                                                                       \
        const int rc_ = min(32, a.cin_p - ck_ * 32);                                                 \
        const bool ok_ = in_ && 8 * (q_ & 3) < rc_;                                                  \
        const unsigned short* s_ = reinterpret_cast<const unsigned short*>(ximg_ + o_) +             \
                                   (ck_ * 64 + (q_ >> 2) * rc_ + 8 * (q_ & 3));                      \
        xreg[k_] = *reinterpret_cast<const f32x4*>(ok_ ? reinterpret_cast<const float*>(s_) : kZeroPage); \
      }                                                                                              \
    }                                                                                                \
    }                                                                                                \
  }
#define WS_STORE_X(K0, K1, BUF)                                                                      \
  _Pragma("unroll") for (int k_ = (K0); k_ < (K1); ++k_) {                                           \
    const int u_ = ptid + k_ * 256;                                                                  \
    const int p_ = u_ >> UPPS, q_ = u_ & (UPP - 1);                                                  \
    if (p_ < npix_lds) {                                                                             \
      f32x4 v_ = xreg[k_];                                                                           \
      if (FIN) {                                                                                     \
        v_[0] = fmaxf(fmaf(v_[0], in_sc[0], in_sh[0]), 0.f);                                         \
        v_[1] = fmaxf(fmaf(v_[1], in_sc[1], in_sh[1]), 0.f);                                         \
        v_[2] = fmaxf(fmaf(v_[2], in_sc[2], in_sh[2]), 0.f);                                         \
        v_[3] = fmaxf(fmaf(v_[3], in_sc[3], in_sh[3]), 0.f);                                         \
      }                                                                                              \
      unsigned char* row_ = xs + (BUF) * XBYTES + p_ * PITCH;                                        \
      const int sx_ = 0;                                                                             \
      if (CVT) {                                                                                     \
        bf16x4 hi_, lo_;                                                                             \
        hi_[0] = (ET)v_[0];                                                                          \
        hi_[1] = (ET)v_[1];                                                                          \
        hi_[2] = (ET)v_[2];                                                                          \
        hi_[3] = (ET)v_[3];                                                                          \
        lo_[0] = (ET)(v_[0] - (float)hi_[0]);                                                        \
        lo_[1] = (ET)(v_[1] - (float)hi_[1]);                                                        \
        lo_[2] = (ET)(v_[2] - (float)hi_[2]);                                                        \
        lo_[3] = (ET)(v_[3] - (float)hi_[3]);                                                        \
        *reinterpret_cast<bf16x4*>(row_ + (((q_ >> 1) ^ sx_) << 4) + (q_ & 1) * 8) = hi_;            \
        if (NP == 3) *reinterpret_cast<bf16x4*>(row_ + ((((q_ >> 1) + 4) ^ sx_) << 4) + (q_ & 1) * 8) = lo_; \
      } else {                                                                                       \
        *reinterpret_cast<f32x4*>(row_ + ((q_ ^ sx_) << 4)) = v_;                                    \
      }                                                                                              \
    }                                                                                                \
  }
    // weights of global phase PH = 3 * stage + r: chunk = stage % nchunks, tap row r
#define WS_LOAD_W(SLOT, PH)                                                                          \
  {                                                                                                  \
    const int phw_ = (PH);                                                                           \
    const int ck_ = ck0 + (phw_ / 3) % nck, r_ = phw_ % 3;                                           \
    _Pragma("unroll") for (int k_ = 0; k_ < WU; ++k_) {                                              \
      const int u_ = min(ptid + k_ * 256, WUNITS - 1);                                               \
      const int t_ = u_ / (NB * 8);                                                                  \
      const int rem_ = u_ - t_ * (NB * 8);                                                           \
      wreg[SLOT][k_] = wpk[(((size_t)ck_ * 9 + (r_ * 3 + t_)) * a.cout_pad + co0 + (rem_ >> 3)) * 8 + (rem_ & 7)]; \
    }                                                                                                \
  }
#define WS_STORE_W(SLOT, BUF)                                                                        \
  _Pragma("unroll") for (int k_ = 0; k_ < WU; ++k_) {                                                \
    const int u_ = ptid + k_ * 256;                                                                  \
    if (u_ < WUNITS) {                                                                               \
      const int t_ = u_ / (NB * 8);                                                                  \
      const int rem_ = u_ - t_ * (NB * 8);                                                           \
      const int wrow_ = t_ * NB + (rem_ >> 3);                                                       \
      *reinterpret_cast<u32x4*>(ws + (BUF) * WROWB + wrow_ * WPITCH + (((rem_ & 7) ^ (WSWZ ? (wrow_ & 7) : 0)) << 4)) = wreg[SLOT][k_]; \
    }                                                                                                \
  }
    // WDMA: weights of phase PH straight into weight buffer BUF (wave-uniform LDS base + lane * 16)
#define WS_DMA_W(BUF, PH)                                                                            \
  {                                                                                                  \
    const int phw_ = (PH);                                                                           \
    const int ck_ = ck0 + (phw_ / 3) % nck, r_ = phw_ % 3;                                           \
    const u32x4* src_ = wpk + ((size_t)(ck_ * 9 + r_ * 3) * a.cout_pad + co0) * 8;                   \
    _Pragma("unroll") for (int k_ = 0; k_ < WU; ++k_) {                                              \
      if (ptid + k_ * 256 < WUNITS) { /* a multiple of 128 units: whole waves */                     \
        /* inline asm, not __builtin_amdgcn_global_load_lds: with a DMA it knows of in flight the compiler waits   \
           vmcnt(0) in front of every use of an ordinary load (the input tile's ds_writes), which cuts the input    \
           prefetch to one phase (measured: +5..10 % on the 256x256 forward layers); unseen DMAs only make its      \
           counted waits more conservative.  M0 = LDS byte address of lane 0, saved and restored in the statement */ \
        const unsigned dst_ = __builtin_amdgcn_readfirstlane(                                        \
            ws_lds + (unsigned)((BUF) * WROWB + (k_ * 256 + (ptid & ~63)) * 16));                    \
        unsigned keep_;                                                                              \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(src_ + wsrc[k_]), "s"(dst_) : "memory");                   \
      }                                                                                              \
    }                                                                                                \
  }
    // all vector-memory operations but the N youngest (this phase's input loads) are done: the DMA'd weights are in LDS
#ifdef MIMO_CONV_STAMPS
    unsigned long long sp_wait = 0;
    const unsigned long long sp_begin = STAMP_NOW();
#define WS_DMA_WAIT(N)                                                                               \
  {                                                                                                  \
    const unsigned long long s0_ = STAMP_NOW();                                                      \
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");    \
    sp_wait += STAMP_NOW() - s0_;                                                                    \
  }
#else
#define WS_DMA_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
#endif
    const int nphases = 3 * nstages;
    WS_LOAD_SS(in_sc, in_sh, 0)
    WS_LOAD_X(0, XU, 0)  // nstages >= 1: the grid never exceeds the tile count
    if (WDMA) {
      WS_DMA_W(0, 0)
    } else {
      WS_LOAD_W(0, 0)
      WS_LOAD_W(1, 1)
      WS_LOAD_W(2, 2)
    }
    WS_STORE_X(0, XU, 0)
    if (!WDMA) {
      WS_STORE_W(0, 0)
    }
    WS_LOAD_SS(in_sc_n, in_sh_n, min(1, nstages - 1))
    WS_LOAD_X(0, XU, min(1, nstages - 1))
    if (WDMA) {
      WS_DMA_WAIT(XU)
    } else {
      __syncthreads();  // phase 0 is staged
    }
    // during phase (j, r): store W(phase + 1) and part r of input stage j + 1 into the buffers the
    // consumers released at the last barrier; then refill those registers two / three phases ahead
    for (int j = 0; j < nstages; ++j) {
      const int xb = (j + 1) & 1;
      if (FIN) { /* the constants of stage j + 1: loaded one stage ago, older than every load still in flight */
        in_sc = in_sc_n;
        in_sh = in_sh_n;
      }
#define WS_PHASE(R)                                                                                  \
  {                                                                                                  \
    /* no conditionals: past the end the loads re-read the last stage / phase and the stores go to   \
       buffers nobody reads any more -- straight-line code lets the compiler keep exact vmcnt counts \
       (a conditional load forces a full drain at the join and collapses the prefetch depth) */      \
    const int ph_ = 3 * j + (R);                                                                     \
    if (WDMA) {                                                                                      \
      if ((R) == 0) {                                                                                \
        WS_LOAD_SS(in_sc_n, in_sh_n, min(j + 2, nstages - 1)) /* in front of this phase's DMA and input loads */ \
      }                                                                                              \
      if (!(abl & 2)) {                                                                              \
        WS_STORE_X((R) * XP, ((R) + 1) * XP, xb)                                                     \
      }                                                                                              \
      if (!(abl & 4)) {                                                                              \
        WS_DMA_W((ph_ + 1) & 1, min(ph_ + 1, nphases - 1))                                           \
      }                                                                                              \
      if (!(abl & 1)) {                                                                              \
        WS_LOAD_X((R) * XP, ((R) + 1) * XP, min(j + 2, nstages - 1))                                 \
        WS_DMA_WAIT(XP)                                                                              \
      } else {                                                                                       \
        WS_DMA_WAIT(0)                                                                               \
      }                                                                                              \
    } else {                                                                                         \
      WS_STORE_W(((R) + 1) % 3, (ph_ + 1) & 1)                                                       \
      WS_LOAD_W((R), min(ph_ + 3, nphases - 1))                                                      \
      WS_STORE_X((R) * XP, ((R) + 1) * XP, xb)                                                       \
      WS_LOAD_X((R) * XP, ((R) + 1) * XP, min(j + 2, nstages - 1))                                   \
      __syncthreads();                                                                               \
    }                                                                                                \
  }
      WS_PHASE(0)
      WS_PHASE(1)
      WS_PHASE(2)
#undef WS_PHASE
    }
#undef WS_LOAD_SS
#undef WS_LOAD_X
#undef WS_STORE_X
#undef WS_LOAD_W
#undef WS_STORE_W
#undef WS_DMA_W
#undef WS_DMA_WAIT
#ifdef MIMO_CONV_STAMPS
    if (ptid == 0) {
      atomicAdd(&g_conv_stamps[(FWD ? 0 : 4) + 2], sp_wait);
      atomicAdd(&g_conv_stamps[(FWD ? 0 : 4) + 3], STAMP_NOW() - sp_begin);
    }
#endif
    if (FWD && a.stats) __syncthreads();  // the consumers combine their BatchNorm sums through LDS (see the end)
    return;
  }

  // =============================== consumers ===============================
#ifdef MIMO_CONV_STAMPS
  unsigned long long st_wait = 0;
  const unsigned long long st_begin = STAMP_NOW();
#define STAMP_BAR_C                                  \
  {                                                  \
    const unsigned long long s0_ = STAMP_NOW();      \
    __syncthreads();                                 \
    st_wait += STAMP_NOW() - s0_;                    \
  }
#else
#define STAMP_BAR_C __syncthreads();
#endif
  // Software pipeline over taps: the 2*(MF+NF) fragment reads of tap t+1 are issued before the 3*MF*NF
  // MFMAs of tap t (two register sets; sched_group_barrier pins the order — left alone, the compiler
  // reads every fragment right before its first use and the lone MFMA wave of the SIMD eats the LDS
  // latency several times per tap).  The pipeline runs across the phase barrier: the last tap of a
  // phase is multiplied after the barrier, under the first reads of the next phase.
  const int lr = lane & 15, g = lane >> 4;
  int pbase[kWsMaxMF];
#pragma unroll
  for (int m = 0; m < MF; ++m) {
    int idx = (wave * MF + m) * 16 + lr;
    if (idx >= npix_out) idx = 0;
    const int r = idx / TC, c = idx - r * TC;
    pbase[m] = (r * TCP + c) * PITCH + g * 16;
  }
  // weight rows: row & 7 == lr & 7 (row = tap * NB + nf * 16 + lr), so the swizzle is a per-lane constant;
  // the lo half is slot + 4, i.e. the hi address with bit 6 flipped
  // phases of a paired chunk (see the kernel comment); second phase: K groups 2-3 meet zero weights and re-read slots
  // 0-1 of their own row (any finite data)
  const int pair_d = g >= 2 ? TCP * PITCH - 32 : 0, pair_d2 = g >= 2 ? -32 : 0;
  const int wbase = WSWZ ? lr * WPITCH + ((g ^ (lr & 7)) << 4) : lr * WPITCH + g * 16;
  const int wlo = WSWZ ? ((wbase ^ 64) - wbase) : 64;
  // Accumulators are kept TRANSPOSED (MFMA called with the weight fragment as A and the pixel fragment as B):
  // lane (lr, g) of fragment (m, nf) holds pixel lr of the fragment and output channels nf*16 + g*4 .. +3, so the
  // epilogue issues one 16-byte store per fragment and one pixel-address computation per m — the
  // pixel-major layout needed four 4-byte stores and four address computations, and on the thin layers
  // (one chunk per tile) that epilogue was 40 % of the kernel (ablation: 235 -> 137 us without it).
  f32x4 bv[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
    bv[nf] = (FWD && a.bias) ? *reinterpret_cast<const f32x4*>(a.bias + co0 + nf * 16 + g * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  const float winv = !F16 ? 1.f : a.wmax ? w16_scale(*a.wmax, true) : 1.f / kF16WeightScale;  // the fp16 image's scale
  (void)winv;
  // tile-relative (row, column) of this lane's pixel in each of its MF fragments; row 0x4000 = not in the tile
  int prc[kWsMaxMF];
#pragma unroll
  for (int m = 0; m < MF; ++m) {
    const int idx = (wave * MF + m) * 16 + lr;
    const int r = idx / TC, c = idx - r * TC;
    prc[m] = idx < npix_out ? (r << 16) | c : (0x4000 << 16);
  }
  const int nphases = 3 * nstages;
  // BatchNorm partial sums: accumulated over all tiles of this (persistent) workgroup, written once —
  // one row per (workgroup, consumer wave) instead of one per (tile, wave)
  f32x4 s1[NF], s2[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    s1[nf] = f32x4{0.f, 0.f, 0.f, 0.f};
    s2[nf] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  f32x4 acc[kWsMaxMF][NF];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[m][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
  // The order is pinned with sched_group_barriers for the data gradient only: measured 7.9 -> 7.3 ms per
  // step there, but 6.9 -> 7.4 ms for the forward, whose producers carry the fp32 -> fp16 split and share
  // the SIMD's vector issue with the MFMAs (a v_mfma_16x16x32 holds it for 8 of its 16 cycles).
#ifdef MIMO_CONV_PIN_FWD
  constexpr bool PINNED = true;  // A/B build: the forward's consumers pinned as well (round 2 measured it slower; round 4 re-checked)
#else
  constexpr bool PINNED = !CVT;
#endif
  constexpr int RA = NP == 3 ? 2 : 1, RB = NF * RA;  // LDS reads per A fragment / per tap's B fragments
  // register sets: A fragments alternate per 16-pixel fragment m, B fragments per tap
  bf16x8 ah[2], al[2], bh[2][NF], bl[2][NF];

#define C_READ_B(BS, KW)                                                                             \
  if (!(abl & 16)) _Pragma("unroll") for (int nf = 0; nf < NF; ++nf) {                                                \
    bh[BS][nf] = *reinterpret_cast<const bf16x8*>(wb_ + ((KW) * NB + nf * 16) * WPITCH);             \
    if (NP == 3) bl[BS][nf] = *reinterpret_cast<const bf16x8*>(wb_ + ((KW) * NB + nf * 16) * WPITCH + wlo); \
  }
  // ro_: tile row of the phase's taps (r_; phases of a paired chunk: 0 and 2); pd_: the per-lane constant of a paired
  // chunk's phases (first: K groups 2-3 read one tile row down, slots 0-1), else 0
#define C_READ_A(AS, KW, M)                                                                     \
  if (!(abl & 16)) {                                                                                                  \
    const unsigned char* p_ = xb_ + pbase[M] + pd_ + (ro_ * TCP + (KW)) * PITCH;                     \
    ah[AS] = *reinterpret_cast<const bf16x8*>(p_);                                                   \
    if (NP == 3) al[AS] = *reinterpret_cast<const bf16x8*>(p_ + 64);                                 \
  }
#define C_MFMA(AS, BS, M)                                                                            \
  if (!(abl & 8)) _Pragma("unroll") for (int nf = 0; nf < NF; ++nf) {                                                \
    if (NP == 3) {                                                                                   \
      acc[M][nf] = mfma16(bh[BS][nf], al[AS], acc[M][nf]);                                           \
      acc[M][nf] = mfma16(bl[BS][nf], ah[AS], acc[M][nf]);                                           \
    }                                                                                                \
    acc[M][nf] = mfma16(bh[BS][nf], ah[AS], acc[M][nf]);                                             \
  }
#define C_PIN(NREADS)                                                                                \
  if (PINNED) {                                                                                      \
    __builtin_amdgcn_sched_group_barrier(0x100, (NREADS), 0); /* DS reads issued ahead */            \
    __builtin_amdgcn_sched_group_barrier(0x008, NP * NF, 0);  /* MFMAs of the current fragment */    \
  }
  // one tap: fragment m+1 (or fragment 0 of the next tap, with that tap's B set) is read under the
  // MFMAs of fragment m; LAST = last tap of the phase (its fragment 3 is multiplied after the barrier)
#define C_TAP(KW, BS, LAST)                                                                          \
  {                                                                                                  \
    C_READ_A(1, KW, 1)                                                                               \
    if (!(LAST)) {                                                                                   \
      C_READ_B((BS) ^ 1, (KW) + 1)                                                                   \
    }                                                                                                \
    C_MFMA(0, BS, 0)                                                                                 \
    C_PIN((LAST) ? RA : RA + RB)                                                                     \
    if (MF == 4) {                                                                                   \
      C_READ_A(0, KW, 2)                                                                             \
      C_MFMA(1, BS, 1)                                                                               \
      C_PIN(RA)                                                                                      \
      C_READ_A(1, KW, 3)                                                                             \
      C_MFMA(0, BS, 2)                                                                               \
      C_PIN(RA)                                                                                      \
    }                                                                                                \
    if (!(LAST)) {                                                                                   \
      C_READ_A(0, (KW) + 1, 0)                                                                       \
      C_MFMA(1, BS, MF - 1)                                                                          \
      C_PIN(RA)                                                                                      \
    }                                                                                                \
  }
  // bias, store, BatchNorm partial sums of tile TI (its accumulators are complete), then clear them
#define C_EPILOGUE(TI)                                                                               \
  {                                                                                                  \
    int t_ = vbx + (TI) * gx;                                                                        \
    const int tx_ = t_ % tilesX;                                                                     \
    t_ /= tilesX;                                                                                    \
    const int ty_ = t_ % tilesY;                                                                     \
    const int n_ = t_ / tilesY;                                                                      \
    const int y0_ = ty_ * TR, x0_ = tx_ * TC;                                                        \
    OT* yimg = reinterpret_cast<OT*>(a.y) + ((size_t)ks * a.N + n_) * a.Ho * a.Wo * a.ldy + co0 + g * 4; \
    if (abl & 32) { /* timing only: keep the accumulators alive, no stores / arithmetic per element */ \
      f32x4 k_ = acc[0][0];                                                                          \
      _Pragma("unroll") for (int m = 0; m < MF; ++m)                                                 \
        _Pragma("unroll") for (int nf = 0; nf < NF; ++nf) k_ = k_ + acc[m][nf];                      \
      if (k_[0] + k_[1] + k_[2] + k_[3] == 12345.678f) yimg[0] = (OT)k_[0];                          \
    }                                                                                                \
    if (!(abl & 32)) _Pragma("unroll") for (int m = 0; m < MF; ++m) {                                \
      const int oy = y0_ + (prc[m] >> 16), ox = x0_ + (prc[m] & 0xffff);                             \
      if (oy < a.Ho && ox < a.Wo) {                                                                  \
        OT* yp = yimg + ((size_t)oy * a.Wo + ox) * a.ldy;                                            \
        _Pragma("unroll") for (int nf = 0; nf < NF; ++nf) {                                          \
          /* bias, inference epilogue and BatchNorm sums exist on the forward only (CVT); the data   \
             gradient stores its accumulators as they are */                                         \
          f32x4 v = acc[m][nf];                                                                      \
          if (F16) v = v * winv;                                                                     \
          if (FWD) v = v + bv[nf];                                                                   \
          const int c_ = co0 + nf * 16 + g * 4;                                                      \
          if (FWD && a.ep_scale) {                                                                   \
            if (a.status && !(isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2]) && isfinite(v[3]))) atomicOr(a.status, 1); \
            const f32x4 esc_ = *reinterpret_cast<const f32x4*>(a.ep_scale + c_);                     \
            const f32x4 esh_ = *reinterpret_cast<const f32x4*>(a.ep_shift + c_);                     \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                       \
              const float mk_ = (a.ep_mask && c_ + i_ < a.ep_mask_ld)                                \
                                    ? a.ep_mask[(size_t)n_ * a.ep_mask_ld + c_ + i_] : 1.f;          \
              v[i_] = fmaxf(fmaf(v[i_], esc_[i_], esh_[i_]), 0.f) * mk_;                             \
            }                                                                                        \
          }                                                                                          \
          if (c_ < a.cout_store) {                                                                   \
            if (OUT16) {                                                                             \
              bf16x4 o_;                                                                             \
              o_[0] = (ET)v[0];                                                                      \
              o_[1] = (ET)v[1];                                                                      \
              o_[2] = (ET)v[2];                                                                      \
              o_[3] = (ET)v[3];                                                                      \
              *reinterpret_cast<bf16x4*>(yp + nf * 16) = o_;                                         \
            } else {                                                                                 \
              *reinterpret_cast<f32x4*>(yp + nf * 16) = v;                                           \
            }                                                                                        \
          }                                                                                          \
          if (FWD) {                                                                                 \
            s1[nf] += v;                                                                             \
            s2[nf] += v * v;                                                                         \
          }                                                                                          \
        }                                                                                            \
      }                                                                                              \
    }                                                                                                \
    _Pragma("unroll") for (int m = 0; m < MF; ++m)                                                   \
      _Pragma("unroll") for (int nf = 0; nf < NF; ++nf) acc[m][nf] = f32x4{0.f, 0.f, 0.f, 0.f};      \
  }
  // one phase (tap row r_ of stage j_).  On entry the previous phase's last fragment is pending: A set 1,
  // B set P; it is multiplied after the barrier, under the first reads of this phase.
#define C_PHASE(P, FIRST)                                                                            \
  {                                                                                                  \
    const int j_ = ph / 3, r_ = ph - 3 * j_;                                                         \
    const unsigned char* xb_ = xs + (j_ & 1) * XBYTES;                                               \
    const unsigned char* wb_ = ws + (ph & 1) * WROWB + wbase;                                        \
    const bool pc_ = PAIR && (j_ + 1) % nchunks == 0; /* phase of a paired chunk (r_ = 0, 1) */       \
    const int ro_ = pc_ ? 2 * r_ : r_;                                                               \
    const int pd_ = pc_ ? (r_ == 0 ? pair_d : pair_d2) : 0;                                          \
    STAMP_BAR_C /* this phase is staged; the buffers of the previous one are released */             \
    C_READ_B((P) ^ 1, 0)                                                                             \
    C_READ_A(0, 0, 0)                                                                                \
    if (!(FIRST)) {                                                                                  \
      C_MFMA(1, P, MF - 1)                                                                           \
      C_PIN(RA + RB)                                                                                 \
      if (tile_done) {                                                                               \
        C_EPILOGUE(ti)                                                                               \
        ++ti;                                                                                        \
      }                                                                                              \
    }                                                                                                \
    C_TAP(0, (P) ^ 1, false)                                                                         \
    C_TAP(1, P, false)                                                                               \
    C_TAP(2, (P) ^ 1, true)                                                                          \
    tile_done = r_ == (PAIR ? 1 : 2) && (j_ + 1) % nck == 0;                                         \
    ++ph;                                                                                            \
    if (PAIR) { /* third phase of a paired chunk: nothing to multiply */                             \
      if (ph < nphases && ph % 3 == 2 && (ph / 3 + 1) % nchunks == 0) {                              \
        __syncthreads();                                                                             \
        ++ph;                                                                                        \
      }                                                                                              \
    }                                                                                                \
  }

  int ph = 0, ti = 0;
  bool tile_done = false;
  // (PAIR: ph also counts the bare-barrier phases, which C_PHASE consumes behind the phase in front of them)
  C_PHASE(1, true)  // leaves B set 0 pending
  while (ph + (PAIR ? 2 : 1) < nphases) {  // at least two phases to go (PAIR: the very last one is a bare barrier)
    C_PHASE(0, false)
    C_PHASE(1, false)
  }
  if (ph < nphases) {  // odd number of remaining phases
    C_PHASE(0, false)
    C_MFMA(1, 1, MF - 1)
  } else {
    C_MFMA(1, 0, MF - 1)
  }
  C_EPILOGUE(ti)
#ifdef MIMO_CONV_STAMPS
  if (tid == 0) {
    atomicAdd(&g_conv_stamps[(FWD ? 0 : 4) + 0], st_wait);
    atomicAdd(&g_conv_stamps[(FWD ? 0 : 4) + 1], STAMP_NOW() - st_begin);
  }
#endif
  __syncthreads();  // matches the producers' last barrier: their (dead) LDS stores are done
  if (FWD && a.stats) {
    // one partial-statistics row per workgroup: the four consumer waves add their sums through LDS (4 x fewer rows
    // for the column reduction that follows)
    float* red = reinterpret_cast<float*>(xs);  // [4 waves][2][NB]
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // sum over the 16 pixel lanes of each channel group
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
          s1[nf][i] += __shfl_xor(s1[nf][i], d);
          s2[nf][i] += __shfl_xor(s2[nf][i], d);
        }
      }
      if (lr == 0) {
        *reinterpret_cast<f32x4*>(red + (wave * 2 + 0) * NB + nf * 16 + g * 4) = s1[nf];
        *reinterpret_cast<f32x4*>(red + (wave * 2 + 1) * NB + nf * 16 + g * 4) = s2[nf];
      }
    }
    __syncthreads();  // second extra barrier, also executed by the producers
    if (tid < 2 * NB) {
      const int which = tid / NB, c = tid - which * NB;
      const float v = (red[(0 * 2 + which) * NB + c] + red[(1 * 2 + which) * NB + c]) +
                      (red[(2 * 2 + which) * NB + c] + red[(3 * 2 + which) * NB + c]);
      a.stats[((size_t)vbx * 2 + which) * a.cout_pad + co0 + c] = v;
    }
  }
#undef C_READ_A
#undef C_READ_B
#undef C_TAP
#undef C_MFMA
#undef C_PIN
#undef C_EPILOGUE
#undef C_PHASE
}

static bool conv_ws_enabled() {
  static const bool on = !(getenv("MIMO_CONV_WS") && atoi(getenv("MIMO_CONV_WS")) == 0);
  return on;
}

int conv3x3_ws_stat_rows(int, int, int) { return 512 * 4; }  // <= 512 persistent workgroups x 4 consumer waves

// Tap pairing of a <= 16-channel last chunk (see conv3x3_ws_kernel): decided per layer from what the dispatch below
// will launch — the packer lays the chunk's weights out for it, the launch selects the PAIR instance (ConvLaunch::pair).
int conv3x3_pair_tail(int mode, int cin_p, int Ho, int Wo) {
  static const bool on = !(getenv("MIMO_CONV_PAIR_TAIL") && atoi(getenv("MIMO_CONV_PAIR_TAIL")) == 0);
  if (!on || mode < 0 || mode > 1 || !conv_ws_enabled() || Ho * Wo < 256) return 0;
  const int tail = cin_p - 32 * (ceil_div(cin_p, 32) - 1);
  return tail <= 16 ? 1 : 0;
}

template <int NF, int MODE, int MF>
static int launch_ws(const ConvLaunch& a, int* rows, hipStream_t stream) {
  int TR, TC;
  pick_tile_n(a.Ho, a.Wo, WsTile<MF>::NPIX, WsTile<MF>::MAXPIX, &TR, &TC);
  const int tilesY = ceil_div(a.Ho, TR), tilesX = ceil_div(a.Wo, TC);
  const int numTiles = a.N * tilesY * tilesX;
  const int coTiles = a.cout_pad / (NF * 16);
  // persistent: one workgroup per CU (two for the 128-pixel / <= 32-channel instances), every workgroup of a
  // launch walks the same number of tiles
  constexpr int kWgPerCU = (MF == 2 && NF <= 2) ? 2 : 1;
  const int ksplit = a.ksplit > 1 ? a.ksplit : 1;
  if (ksplit > 1) {
    const int nchunks = ceil_div(a.cin_p, 32), cpk = ceil_div(nchunks, ksplit);
    if (MODE > 1 || a.pair || a.bias || a.stats || a.ep_scale || a.cin_p % 32 != 0 || cpk * (ksplit - 1) >= nchunks) {
      set_error("conv3x3 split: K split %d needs a split16 launch without bias / stats / epilogue, whole 32-channel chunks "
                "and at least one chunk per split (cin_p %d)", ksplit, a.cin_p);
      return MIMO_ERR_INVALID;
    }
  }
  int gx = max(1, 256 * kWgPerCU / (coTiles * ksplit));
  if (gx > numTiles) gx = numTiles;
  const int per = ceil_div(numTiles, gx);
  gx = ceil_div(numTiles, per);
  if (rows) *rows = gx;  // one partial-statistics row per workgroup
  dim3 grid(gx * coTiles * ksplit);
  static const int xcd_ = !(getenv("MIMO_CONV_XCD_ORDER") && atoi(getenv("MIMO_CONV_XCD_ORDER")) == 0);
  // MIMO_CONV_WDMA=0: weights staged through registers (ds_write) as before
  static const bool wdma = !(getenv("MIMO_CONV_WDMA") && atoi(getenv("MIMO_CONV_WDMA")) == 0);
  const int xcd = xcd_;
  if (a.in_scale) {
    if constexpr (MODE == 1) {
      if (!wdma || a.ep_scale || !a.in_shift || (a.pair && a.pair != conv3x3_pair_tail(MODE, a.cin_p, a.Ho, a.Wo))) {
        set_error("conv3x3 split: the input BatchNorm + ReLU can be fused into the split16 training forward (weights by DMA) only");
        return MIMO_ERR_INVALID;
      }
      if (a.pair)
        hipLaunchKernelGGL((conv3x3_ws_kernel<NF, MODE, MF, true, true, true>), grid, dim3(512), 0, stream, a, TR, TC, tilesY, tilesX, numTiles, xcd, gx, coTiles);
      else
        hipLaunchKernelGGL((conv3x3_ws_kernel<NF, MODE, MF, false, true, true>), grid, dim3(512), 0, stream, a, TR, TC, tilesY, tilesX, numTiles, xcd, gx, coTiles);
      MIMO_KERNEL_CHECK();
      return MIMO_OK;
    } else {
      set_error("conv3x3 split: the input BatchNorm + ReLU can be fused into the split16 forward only");
      return MIMO_ERR_INVALID;
    }
  }
  if (a.pair) {
    if constexpr (MODE <= 1) {
      if (a.pair != conv3x3_pair_tail(MODE, a.cin_p, a.Ho, a.Wo)) {
        set_error("conv3x3 split: weights packed for tap pairing, launch is not");
        return MIMO_ERR_INVALID;
      }
      hipLaunchKernelGGL((conv3x3_ws_kernel<NF, MODE, MF, true>), grid, dim3(512), 0, stream, a, TR, TC, tilesY, tilesX, numTiles, xcd, gx, coTiles);
    } else {
      set_error("conv3x3 split: tap pairing exists for the split16 modes only");
      return MIMO_ERR_INVALID;
    }
  } else if (wdma)
    hipLaunchKernelGGL((conv3x3_ws_kernel<NF, MODE, MF>), grid, dim3(512), 0, stream, a, TR, TC, tilesY, tilesX, numTiles, xcd, gx, coTiles);
  else
    hipLaunchKernelGGL((conv3x3_ws_kernel<NF, MODE, MF, false, false>), grid, dim3(512), 0, stream, a, TR, TC, tilesY, tilesX, numTiles, xcd, gx, coTiles);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

template <int MF, int NF, int MODE>
static int launch_bf16x3(const ConvLaunch& a, int* rows, hipStream_t stream) {
  int TR, TC;
  pick_tile_n(a.Ho, a.Wo, TileCfg<MF>::NPIX, TileCfg<MF>::MAXPIX, &TR, &TC);
  const int tilesY = ceil_div(a.Ho, TR), tilesX = ceil_div(a.Wo, TC);
  if (rows) *rows = a.N * tilesY * tilesX;
  dim3 grid(a.N * tilesY * tilesX, a.cout_pad / (NF * 16));
  hipLaunchKernelGGL((conv3x3_bf16x3_kernel<MF, NF, MODE>), grid, dim3(512), 0, stream, a, TR, TC, tilesY, tilesX);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// MF = 4 (512-pixel tiles) when the image is large enough to fill them, else 2.
static bool use_big_tile(int Ho, int Wo) { return Ho * Wo >= 1024; }

// 1 when the launch runs on a kernel whose loaders can apply ConvLaunch::in_scale / in_shift (the wide kernel and the
// 256-pixel wave-specialised kernel, split16 forward)
int conv3x3_split_fuses_input(int mode, int wide, int Ho, int Wo) {
  static const bool wdma = !(getenv("MIMO_CONV_WDMA") && atoi(getenv("MIMO_CONV_WDMA")) == 0);
  return mode == 1 && (wide != 0 || (conv_ws_enabled() && wdma && Ho * Wo >= 256)) ? 1 : 0;
}

template <int MODE>
static int conv3x3_split_dispatch(const ConvLaunch& a, int* rows, hipStream_t stream) {
  if (a.in_scale && !conv3x3_split_fuses_input(MODE, 0, a.Ho, a.Wo)) {
    set_error("conv3x3 split: this launch cannot apply the input BatchNorm + ReLU in its loader (conv3x3_split_fuses_input)");
    return MIMO_ERR_INVALID;
  }
  const int nfr = a.cout_pad / 16;
  int nf = 4;
  while (nfr % nf != 0) --nf;
  if (conv_ws_enabled() && a.Ho * a.Wo >= 256) {
    // Channel-tile width of the persistent kernel by cost, not just "the widest that divides": a launch with fewer
    // (pixel tile, channel tile) pairs than CUs (the 16x16 .. 64x64 layers at a few images per GPU — the per-GPU
    // batch of a strong-scaling run) finishes sooner with narrow channel tiles on many CUs than with wide ones on a
    // few: time ~ (tiles per workgroup) x (per-phase fixed cost + NF x MFMA time), fixed cost ~ one NF unit.
    static const bool adapt = !(getenv("MIMO_CONV_ADAPTIVE_NF") && atoi(getenv("MIMO_CONV_ADAPTIVE_NF")) == 0);
    if (adapt && a.ksplit <= 1) {  // (a K split keeps the widest channel tile: conv3x3_ksplit priced it that way)
      int TR, TC;
      pick_tile_n(a.Ho, a.Wo, WsTile<4>::NPIX, WsTile<4>::MAXPIX, &TR, &TC);
      const int numTiles = a.N * ceil_div(a.Ho, TR) * ceil_div(a.Wo, TC);
      long best = -1;
      for (int c = 4; c >= 1; --c) {
        if (nfr % c) continue;
        const int gx = max(1, min(numTiles, 256 / (nfr / c)));
        const long cost = (long)ceil_div(numTiles, gx) * (1 + c);
        if (best < 0 || cost < best) best = cost, nf = c;
      }
    }
  }
  // NF <= 2: little MFMA work per tile -> 256-pixel tiles so that two workgroups share a CU and
  // one's loads / stores overlap the other's MFMAs
  if (conv_ws_enabled() && a.Ho * a.Wo >= 256) {
    // 128-pixel tiles / two workgroups per CU: forward only (measured per layer on one box: forward 30->30 at
    // 256x256 169 -> 147 us, 45->30 282 -> 253 us; the data gradient of the same shapes 132 -> 145 us)
    static const bool mf2_on = !(getenv("MIMO_CONV_WS_MF2") && atoi(getenv("MIMO_CONV_WS_MF2")) == 0);
    const bool mf2 = mf2_on && (MODE == 1 || MODE == 2 || MODE == 4 || MODE == 6);
    switch (nf) {
      case 4: return launch_ws<4, MODE, 4>(a, rows, stream);
      case 3: return launch_ws<3, MODE, 4>(a, rows, stream);
      case 2: return mf2 ? launch_ws<2, MODE, 2>(a, rows, stream) : launch_ws<2, MODE, 4>(a, rows, stream);
      default: return mf2 ? launch_ws<1, MODE, 2>(a, rows, stream) : launch_ws<1, MODE, 4>(a, rows, stream);
    }
  }
  if (a.pair) {
    set_error("conv3x3 split: weights packed for tap pairing, launch is not");
    return MIMO_ERR_INVALID;
  }
  if (use_big_tile(a.Ho, a.Wo) && nf >= 3) {
    switch (nf) {
      case 4: return launch_bf16x3<4, 4, MODE>(a, rows, stream);
      case 3: return launch_bf16x3<4, 3, MODE>(a, rows, stream);
      case 2: return launch_bf16x3<4, 2, MODE>(a, rows, stream);
      default: return launch_bf16x3<4, 1, MODE>(a, rows, stream);
    }
  }
  switch (nf) {
    case 4: return launch_bf16x3<2, 4, MODE>(a, rows, stream);
    case 3: return launch_bf16x3<2, 3, MODE>(a, rows, stream);
    case 2: return launch_bf16x3<2, 2, MODE>(a, rows, stream);
    default: return launch_bf16x3<2, 1, MODE>(a, rows, stream);
  }
}

// ---- K split (round 6) ------------------------------------------------------------------------------------------
// Cost in the units of the channel-tile rule above: a phase of a persistent workgroup costs (1 + NF) — one unit of fixed
// cost (barrier, pipeline latency) and one per 16-channel fragment of MFMA work.  Unsplit: the rule's own optimum over NF.
// Split ks-fold: the widest channel tile, ceil(nchunks / ks) chunks per workgroup, plus the reduction pass (priced at
// kKsplitReduceUnits: one more launch over ks + 1 small tensors).  Taken when it wins by >= 15 %.
constexpr int kKsplitReduceUnits = 14;
int conv3x3_ksplit(int mode, int N, int cin_p, int cout_pad, int Ho, int Wo) {
  static const int force = [] { const char* e = getenv("MIMO_CONV_KSPLIT"); return e ? atoi(e) : -1; }();
  static const bool wdma = !(getenv("MIMO_CONV_WDMA") && atoi(getenv("MIMO_CONV_WDMA")) == 0);
  if (force == 0 || (mode != 0 && mode != 1) || !conv_ws_enabled() || !wdma || Ho * Wo < 256 || cin_p % 32 != 0 ||
      cout_pad % 16 != 0)
    return 1;
  const int nchunks = cin_p / 32;
  if (nchunks < 2) return 1;
  auto legal = [&](int ks) { return ks >= 2 && ceil_div(nchunks, ks) * (ks - 1) < nchunks; };
  if (force >= 2) {
    int ks = min(force, nchunks);
    while (ks >= 2 && !legal(ks)) --ks;
    return ks >= 2 ? ks : 1;
  }
  int TR, TC;
  pick_tile_n(Ho, Wo, WsTile<4>::NPIX, WsTile<4>::MAXPIX, &TR, &TC);
  const int numTiles = N * ceil_div(Ho, TR) * ceil_div(Wo, TC);
  const int nfr = cout_pad / 16;
  long unsplit = -1;
  for (int c = 4; c >= 1; --c) {
    if (nfr % c) continue;
    const int gx = max(1, min(numTiles, 256 / (nfr / c)));
    const long cost = (long)ceil_div(numTiles, gx) * (1 + c) * nchunks * 3;
    if (unsplit < 0 || cost < unsplit) unsplit = cost;
  }
  int nf = 4;
  while (nfr % nf != 0) --nf;
  const int coTiles = nfr / nf;
  int best = 1;
  long best_cost = -1;
  for (int ks = 2; ks <= 8 && ks <= nchunks; ++ks) {
    if (!legal(ks) || coTiles * ks > 256) continue;
    const int gx = max(1, min(numTiles, 256 / (coTiles * ks)));
    const long cost = (long)ceil_div(numTiles, gx) * (1 + nf) * ceil_div(nchunks, ks) * 3 + kKsplitReduceUnits;
    if (best_cost < 0 || cost < best_cost) best_cost = cost, best = ks;
  }
  return (best_cost >= 0 && best_cost * 100 < unsplit * 85) ? best : 1;
}

size_t conv3x3_ksplit_scratch(int mode, int N, int cin_p, int cout_pad, int Ho, int Wo, int ldy) {
  const int ks = conv3x3_ksplit(mode, N, cin_p, cout_pad, Ho, Wo);
  return ks > 1 ? (size_t)ks * N * Ho * Wo * ldy : 0;
}

// out[p][c] = bias[c] + sum over the slabs of part[k][p][c]; stats != nullptr: one row [2][cout_pad] of (sum, sum of squares)
// per workgroup, the layout of the convolution epilogue's BatchNorm partial rows.  256 threads = QB channel quads x PPI pixels.
__global__ __launch_bounds__(256) void conv_ksplit_reduce_kernel(const float* __restrict__ part, int ksplit, int P, int ld, int Cv,
                                                                 const float* __restrict__ bias, float* __restrict__ out,
                                                                 int ldo, float* __restrict__ stats, int cout_pad) {
  __shared__ f32x4 red[2][256];
  const int QB = Cv < 256 ? Cv : 256, PPI = 256 / QB;
  const int ql = threadIdx.x % QB, pl = threadIdx.x / QB;
  const int q = blockIdx.y * QB + ql;
  const bool active = pl < PPI && q < Cv;
  f32x4 s1 = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = s1;
  if (active) {
    const f32x4 b = bias ? *reinterpret_cast<const f32x4*>(bias + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    const size_t slab = (size_t)P * ld;
    for (int p = blockIdx.x * PPI + pl; p < P; p += gridDim.x * PPI) {
      const float* src = part + (size_t)p * ld + 4 * q;
      f32x4 v = *reinterpret_cast<const f32x4*>(src);
      for (int k = 1; k < ksplit; ++k) v += *reinterpret_cast<const f32x4*>(src + k * slab);
      v += b;
      *reinterpret_cast<f32x4*>(out + (size_t)p * ldo + 4 * q) = v;
      s1 += v;
      s2 += v * v;
    }
  }
  if (!stats) return;
  red[0][threadIdx.x] = s1;
  red[1][threadIdx.x] = s2;
  __syncthreads();
  if (pl == 0 && q < Cv) {
    for (int i = 1; i < PPI; ++i) {
      s1 += red[0][i * QB + ql];
      s2 += red[1][i * QB + ql];
    }
    float* row = stats + (size_t)blockIdx.x * 2 * cout_pad;
    *reinterpret_cast<f32x4*>(row + 4 * q) = s1;
    *reinterpret_cast<f32x4*>(row + cout_pad + 4 * q) = s2;
  }
}

int conv3x3_bf16x3_launch_k(const ConvLaunch& a, int mode, int* rows, hipStream_t stream, float* kpart, size_t kpart_floats) {
  const int ks = (a.wide || a.ep_scale || !kpart) ? 1 : conv3x3_ksplit(mode, a.N, a.cin_p, a.cout_pad, a.Ho, a.Wo);
  const size_t need = (size_t)ks * a.N * a.Ho * a.Wo * a.ldy;
  if (ks <= 1 || a.pair || need > kpart_floats || a.cout_store % 4 != 0 || a.ldy % 4 != 0 || (size_t)a.N * a.Ho * a.Wo > (1u << 30))
    return conv3x3_bf16x3_launch(a, mode, rows, stream);
  ConvLaunch p = a;
  p.y = kpart;
  p.bias = nullptr;
  p.stats = nullptr;
  p.ksplit = ks;
  MIMO_TRY(conv3x3_bf16x3_launch(p, mode, nullptr, stream));
  const int P = a.N * a.Ho * a.Wo, Cv = a.cout_store / 4;
  const int QB = Cv < 256 ? Cv : 256, PPI = 256 / QB;
  const int gx = max(1, min(ceil_div(P, PPI * 2), 256));  // <= 256 partial rows, >= 2 pixels per thread
  if (rows) *rows = gx;
  hipLaunchKernelGGL(conv_ksplit_reduce_kernel, dim3(gx, ceil_div(Cv, QB)), dim3(256), 0, stream, kpart, ks, P, a.ldy, Cv, a.bias,
                     a.y, a.ldy, a.stats, a.cout_pad);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// mode: see MIMO_CONV_MODE_CONSTANTS (0 split16 dgrad, 1 split16 forward, 2 bf16 forward, 3 bf16 dgrad)
int conv3x3_bf16x3_launch(const ConvLaunch& a, int mode, int* rows, hipStream_t stream) {
  if (a.wide) return conv3x3_wide_launch(a, mode, rows, stream);  // weights packed for conv_wide.hip
  if (!a.wpk || a.ldx % 4 != 0 || a.cout_pad % 16 != 0 || a.Hi < 2 || a.Wi < 2 || mode < 0 || mode > 7 ||
      (mode >= 4 && (a.ldx % 8 != 0 || a.cin_p % 8 != 0 || a.ldy % 4 != 0))) {
    set_error("conv3x3 split: bad geometry or mode");
    return MIMO_ERR_INVALID;
  }
  switch (mode) {
    case 0: return conv3x3_split_dispatch<0>(a, rows, stream);
    case 1: return conv3x3_split_dispatch<1>(a, rows, stream);
    case 2: return conv3x3_split_dispatch<2>(a, rows, stream);
    case 3: return conv3x3_split_dispatch<3>(a, rows, stream);
    case 4: return conv3x3_split_dispatch<4>(a, rows, stream);
    case 5: return conv3x3_split_dispatch<5>(a, rows, stream);
    case 6: return conv3x3_split_dispatch<6>(a, rows, stream);
    default: return conv3x3_split_dispatch<7>(a, rows, stream);
  }
}

// ---------------------------------------------------------------------------------------
// weight packing for the split kernel: torch OIHW -> [chunk][tap][row][hi 32 | lo 32] bf16
// (row/col maps and the transposed flag as in pack_weights_kernel, conv3x3.hip)
// ---------------------------------------------------------------------------------------
// paired last chunk (conv3x3_pair_tail): tap (row, kw) of channel k < 16 -> of the chunk's nine weight slots, slot kw
// for rows 0 (K lane k) and 1 (K lane 16 + k), slot 3 + kw for row 2 (K lane k).  K lanes 16-31 of slots 3-5 and
// slots 6-8 are never written: they keep the zeros of the allocation.
__device__ __forceinline__ void pair_slot(int tap, int k, int* slot, int* kk) {
  const int row = tap / 3, kw = tap - 3 * row;
  *slot = row == 2 ? 3 + kw : kw;
  *kk = k + (row == 1 ? 16 : 0);
}

template <bool F16>
__global__ void pack_weights_bf16x3_kernel(const float* __restrict__ w, typename Elem<F16>::T* __restrict__ dst, int cout,
                                           int cin, int rows_pad, int cols, int nchunks, const int* __restrict__ row_map,
                                           const int* __restrict__ col_map, int transposed, int pair) {
  typedef typename Elem<F16>::T ET;
  const int total = nchunks * 9 * rows_pad * 32;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int k = i & 31;
    int rest = i >> 5;
    const int row = rest % rows_pad;
    rest /= rows_pad;
    const int tap = rest % 9, chunk = rest / 9;
    const int col = chunk * 32 + k;
    float v = 0.f;
    if (col < cols) {
      const int rm = row_map[row], cm = col_map[col];
      if (rm >= 0 && cm >= 0) {
        const int co = transposed ? cm : rm, ci = transposed ? rm : cm;
        const int kh = transposed ? 2 - tap / 3 : tap / 3, kw = transposed ? 2 - tap % 3 : tap % 3;
        v = w[(((size_t)co * cin + ci) * 3 + kh) * 3 + kw];
      }
    }
    if (F16) v *= kF16WeightScale;
    const ET hi = (ET)v;
    const ET lo = (ET)(v - (float)hi);
    int slot = tap, kk = k;
    if (pair && chunk == nchunks - 1) {
      if (k >= 16) continue;  // K lanes 16-31 of the paired image belong to the odd taps of channels 0-15
      pair_slot(tap, k, &slot, &kk);
    }
    ET* d = dst + (((size_t)chunk * 9 + slot) * rows_pad + row) * 64;
    d[kk] = hi;
    d[32 + kk] = lo;
  }
}

// one launch for every repack of a step (PackJob, common.h); element formulas as in the kernels above / in
// pack_weights_kernel (conv3x3.hip)
__global__ void pack_jobs_kernel(const PackJob* __restrict__ jobs, const float* __restrict__ params) {
  const PackJob j = jobs[blockIdx.y];
  const float* w = params + j.w_off;
  const float wscale = j.wmax ? w16_scale(*j.wmax, false) : kF16WeightScale;  // fp16 kinds (1, 3, 5)
  if (blockIdx.x == 0 && j.bias_n > 0)
    for (int i = threadIdx.x; i < j.bias_n; i += blockDim.x) j.bias_dst[i] = params[j.bias_off + i];
  // One thread per (row, column) = one (output channel, input channel) pair: its nine taps are 36 contiguous bytes
  // of the OIHW tensor, so a wave reads one contiguous span (the tap-major order of round 1 read 4 of every 36
  // bytes per pass: 113 us per step for 60 MB of weights); the nine stores per thread are coalesced across the wave
  // (consecutive columns of one tap).
  const int per_tap = j.total / 9;  // elements of one tap: kind 0: rows_pad * cols; else chunks * rows_pad * 32
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per_tap; i += gridDim.x * blockDim.x) {
    int row, col, k = 0, chunk = 0;
    if (j.kind == 0) {
      col = i % j.cols;
      row = i / j.cols;
    } else if (j.kind >= 5) {  // wide layout, 16-bit storage modes: 32-channel chunks of single values
      k = i & 31;
      const int rest = i >> 5;
      row = rest % j.rows_pad;
      chunk = rest / j.rows_pad;
      col = chunk * 32 + k;
    } else if (j.kind >= 3) {  // wide layout: 16-channel chunks
      k = i & 15;
      const int rest = i >> 4;
      row = rest % j.rows_pad;
      chunk = rest / j.rows_pad;
      col = chunk * 16 + k;
    } else {
      k = i & 31;
      const int rest = i >> 5;
      row = rest % j.rows_pad;
      chunk = rest / j.rows_pad;
      col = chunk * 32 + k;
    }
    float v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) v[t] = 0.f;
    if (col < j.cols && (j.kind < 3 || row < j.map_rows)) {
      const int rm = j.row_map[row], cm = j.col_map[col];
      if (rm >= 0 && cm >= 0) {
        const int co = j.transposed ? cm : rm, ci = j.transposed ? rm : cm;
        const float* src = w + ((size_t)co * j.cin + ci) * 9;
#pragma unroll
        for (int t = 0; t < 9; ++t) v[t] = src[j.transposed ? 8 - t : t];  // transposed: taps flipped (kh, kw -> 2-kh, 2-kw)
      }
    }
    const bool paired = (j.kind == 1 || j.kind == 2) && j.pair && chunk == (j.cols + 31) / 32 - 1;
    if (paired && k >= 16) continue;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float x = v[tap];
      int slot = tap, kk = k;
      if (paired) pair_slot(tap, k, &slot, &kk);
      if (j.kind == 0) {
        reinterpret_cast<float*>(j.dst)[((size_t)tap * j.rows_pad + row) * j.cols + col] = x;
      } else if (j.kind == 5) {
        reinterpret_cast<_Float16*>(j.dst)[(((size_t)chunk * 9 + tap) * j.rows_pad + row) * 32 + k] = (_Float16)(x * wscale);
      } else if (j.kind == 6) {
        reinterpret_cast<__bf16*>(j.dst)[(((size_t)chunk * 9 + tap) * j.rows_pad + row) * 32 + k] = (__bf16)x;
      } else if (j.kind == 3) {
        const float xs = x * wscale;
        const _Float16 hi = (_Float16)xs, lo = (_Float16)(xs - (float)hi);
        _Float16* d = reinterpret_cast<_Float16*>(j.dst) + (((size_t)chunk * 9 + tap) * j.rows_pad + row) * 32;
        d[k] = hi;
        d[16 + k] = lo;
      } else if (j.kind == 4) {
        const __bf16 hi = (__bf16)x, lo = (__bf16)(x - (float)hi);
        __bf16* d = reinterpret_cast<__bf16*>(j.dst) + (((size_t)chunk * 9 + tap) * j.rows_pad + row) * 32;
        d[k] = hi;
        d[16 + k] = lo;
      } else if (j.kind == 1) {
        const float xs = x * wscale;
        const _Float16 hi = (_Float16)xs, lo = (_Float16)(xs - (float)hi);
        _Float16* d = reinterpret_cast<_Float16*>(j.dst) + (((size_t)chunk * 9 + slot) * j.rows_pad + row) * 64;
        d[kk] = hi;
        d[32 + kk] = lo;
      } else {
        const __bf16 hi = (__bf16)x, lo = (__bf16)(x - (float)hi);
        __bf16* d = reinterpret_cast<__bf16*>(j.dst) + (((size_t)chunk * 9 + slot) * j.rows_pad + row) * 64;
        d[kk] = hi;
        d[32 + kk] = lo;
      }
    }
  }
}

// max |w| of each job's weight tensor into its wmax word (jobs without one: nothing).  A workgroup whose maximum stays below
// 64 issues no atomic (the scale only changes from 128 up, w16_scale): for ordinary weights this is one read of the parameters.
__global__ void wabsmax_jobs_kernel(const PackJob* __restrict__ jobs, const float* __restrict__ params, int* __restrict__ status) {
  const PackJob j = jobs[blockIdx.y];
  if (!j.wmax) return;
  const float4* w4 = reinterpret_cast<const float4*>(params + j.w_off);  // (tensor offsets are multiples of 4 floats)
  const int n = j.cout * j.cin * 9, n4 = n / 4;
  float m = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
    const float4 v = w4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (blockIdx.x == 0 && threadIdx.x < n - 4 * n4) m = fmaxf(m, fabsf(params[j.w_off + 4 * n4 + threadIdx.x]));
  __shared__ float wm[4];
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) m = fmaxf(m, __shfl_xor(m, d));
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    // (fmaxf drops a NaN operand: an infinite or NaN weight shows as a maximum that is not an ordinary number)
    if (status && !(m <= 3.0e38f)) atomicOr(status, 1);
    if (m >= 64.f) atomicMax(j.wmax, __float_as_uint(m));  // (non-negative floats order like their bit patterns)
  }
}

int wabsmax_jobs_launch(const PackJob* jobs_dev, int njobs, int max_total, const float* params, hipStream_t stream, int* status) {
  if (njobs <= 0) return MIMO_OK;
  const int gx = max(1, min(ceil_div(max_total / 9, 256 * 8), 128));
  hipLaunchKernelGGL(wabsmax_jobs_kernel, dim3(gx, njobs), dim3(256), 0, stream, jobs_dev, params, status);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

int pack_jobs_launch(const PackJob* jobs_dev, int njobs, int max_total, const float* params, hipStream_t stream) {
  if (njobs <= 0) return MIMO_OK;
  const int gx = max(1, min(ceil_div(max_total / 9, 256 * 2), 512));
  hipLaunchKernelGGL(pack_jobs_kernel, dim3(gx, njobs), dim3(256), 0, stream, jobs_dev, params);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

#ifdef MIMO_CONV_STAMPS
}  // namespace mimo
extern "C" int mimo_debug_conv_stamps(unsigned long long* out) {  // reads and clears the counters
  unsigned long long z[8] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mimo::g_conv_stamps), sizeof(z)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(mimo::g_conv_stamps), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
namespace mimo {
#endif

int pack_weights_bf16x3_launch(const float* w, void* dst, int f16, int cout, int cin, int rows_pad, int cols,
                               const int* row_map, const int* col_map, int transposed, hipStream_t stream, int pair) {
  const int nchunks = ceil_div(cols, 32);
  const int total = nchunks * 9 * rows_pad * 32;
  const int blocks = min(ceil_div(total, 256), 4096);
  if (f16)
    hipLaunchKernelGGL(pack_weights_bf16x3_kernel<true>, dim3(blocks), dim3(256), 0, stream, w,
                       reinterpret_cast<_Float16*>(dst), cout, cin, rows_pad, cols, nchunks, row_map, col_map, transposed, pair);
  else
    hipLaunchKernelGGL(pack_weights_bf16x3_kernel<false>, dim3(blocks), dim3(256), 0, stream, w,
                       reinterpret_cast<__bf16*>(dst), cout, cin, rows_pad, cols, nchunks, row_map, col_map, transposed, pair);
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

}  // namespace mimo
