// Weight gradient of the 3x3 convolution on the bf16 MFMA with split operands (see
// conv_bf16x3.hip for the arithmetic: a = hi + lo, three MFMAs per product block, fp32
// accumulate, ~1e-5 per product — harmless on the gradient side).
//
//   dW[tap][ci][co] = sum_pixels A(pixel + tap, ci) * dz(pixel, co)      (components.py:23,26 autograd)
//
// dz arrives PRE-SPLIT from the BatchNorm-backward kernel — per pixel and 32-channel chunk
// [hi 32 | lo 32] bf16 (elementwise.hip st_split4) — so its staging is a 16-byte copy; the activation operand A is fp32 in
// HBM (it also feeds the fp16-pair forward and the bandwidth-class kernels) and is split on the way in.
//
// GEMM view: M = input channels, N = output channels, K = pixels (32 per v_mfma_f32_16x16x32_bf16).
// Both operands are stored in LDS the way they sit in HBM — pixel-major NHWC rows, split into
// [hi C x bf16 | lo C x bf16] — and are read with ds_read_b64_tr_b16, the hardware transposed
// read, which hands every lane 4 consecutive K (pixels) of its own channel: a tap shift is then
// a whole-row offset (always aligned), with no transposing write pass.
//   * LDS row pitch = 32*m bytes with m = 2 (mod 4) and rows whose index has bit 3 set shifted
//     by one 32-byte block: the 8 rows a half-wave touches (R..R+3, R+8..R+11) fall on 8 distinct
//     bank octets -> conflict-free transposed reads at any tap offset
//   * workgroup = 4 waves, one per SIMD (__launch_bounds__(256,1): the full 512-register file per
//     lane): each wave owns a 32x32 (ci,co) tile for all 9 taps (144 accumulator registers) and the
//     next spatial tile is prefetched into registers while the current one is multiplied
//   * waves are arranged WM x WN over (ci,co) and WK over K (tile rows); WK > 1 partial sums
//     are combined through LDS once, at the end
#include <cstdlib>

#include "common.h"

#ifndef MIMO_WGRAD_NP2_DEPTH
// taps the activation fragments of the two-MFMA weight gradient are read ahead of their MFMAs (a tap is 2 * NI MFMAs = 128
// matrix-pipe cycles at NI = 4, about the latency of a transposed LDS read)
#define MIMO_WGRAD_NP2_DEPTH 1
#endif
#ifndef MIMO_WGRAD_PIN_PROLOGUE
// 1: pin the prologue's LDS reads into a scheduling group of their own, which puts the consumers' fragment reads really
// one tap ahead of their MFMAs in the ISA.  Measured SLOWER (round 4, profiles/r04/wgrad_read_pipeline.txt: +3.5 % per
// layer, wgrad class 6.16-6.29 -> 6.29-6.41 ms in four alternating step pairs): off.
#define MIMO_WGRAD_PIN_PROLOGUE 0
#endif

namespace mimo {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

constexpr int kWgTR = 4, kWgTC = 32, kWgTCP = 34;
constexpr int kWgAPix = (kWgTR + 2) * kWgTCP;  // 204 halo-tile pixels
constexpr int kWgDPix = kWgTR * kWgTC;         // 128 pixels

// smallest pitch = 32*m bytes, m = 2 (mod 4), holding 4*C data bytes + one 32-byte shift block
constexpr int wg_pitch(int C) {
  int m = (4 * C + 32 + 31) / 32;
  while (m % 4 != 2) ++m;
  return 32 * m;
}
// the same for an operand stored as ONE 16-bit plane (16-bit storage modes: no lo half): 2*C data bytes
constexpr int wg_pitch16(int C) {
  int m = (2 * C + 32 + 31) / 32;
  while (m % 4 != 2) ++m;
  return 32 * m;
}

// Operand storage of the weight-gradient kernels (template parameter WM):
//   0  activations fp32 (split into bf16 pairs / rounded to bf16 on the way in), dz pre-split bf16 (hi, lo) records
//   1 / 2  16-bit storage modes: activations AND dz plain NHWC bf16 (1) / fp16 (2), copied as they are
//   3 / 4  the image convolution in those modes: activations fp32 (the packed input image), dz plain bf16 / fp16
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_w;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_w;
template <int WM>
__device__ __forceinline__ f32x4 wg_mfma(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (WM == 2 || WM == 4)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_w, a), __builtin_bit_cast(f16x8_w, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// four fp32 values -> four 16-bit values of the mode's element type (as raw bf16x4 bits)
template <int WM>
__device__ __forceinline__ bf16x4 wg_round4(f32x4 v) {
  if constexpr (WM == 2 || WM == 4) {
    f16x4_w h;
    h[0] = (_Float16)v[0];
    h[1] = (_Float16)v[1];
    h[2] = (_Float16)v[2];
    h[3] = (_Float16)v[3];
    return __builtin_bit_cast(bf16x4, h);
  } else {
    bf16x4 h;
    h[0] = (__bf16)v[0];
    h[1] = (__bf16)v[1];
    h[2] = (__bf16)v[2];
    h[3] = (__bf16)v[3];
    return h;
  }
}

__device__ __forceinline__ bf16x8 tr_read8(const unsigned char* p0, const unsigned char* p1) {
  const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p0);
  const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p1);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// NP == 2 kernels: a bf16 (hi, lo) record pair of 8 channels -> the SAME pair as fp16 values, scaled by `mul` (a power of two):
// hi * mul and lo * mul each have 8 significant bits, so both conversions are exact (no re-splitting, any rounding mode) for
// every element within 2^-21 of the layer's largest |dz| — below that the lo part runs into fp16's subnormals and loses bits
// of a value that contributes nothing.  5 vector instructions per element (re-splitting hi + lo: 8).  Element order kept.
__device__ __forceinline__ void wg_dz_pair_to_f16(f32x4 hi_raw, f32x4 lo_raw, float mul, f32x4* out_hi, f32x4* out_lo) {
  typedef unsigned int u32x4_w __attribute__((ext_vector_type(4)));
  const u32x4_w h = __builtin_bit_cast(u32x4_w, hi_raw), l = __builtin_bit_cast(u32x4_w, lo_raw);
  u32x4_w oh, ol;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    oh[e] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(h[e] << 16) * mul,
                                                                        __uint_as_float(h[e] & 0xffff0000u) * mul));
    ol[e] = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pkrtz(__uint_as_float(l[e] << 16) * mul,
                                                                        __uint_as_float(l[e] & 0xffff0000u) * mul));
  }
  *out_hi = __builtin_bit_cast(f32x4, oh);
  *out_lo = __builtin_bit_cast(f32x4, ol);
}

template <int MI, int NI, int WM, int WN, int WK, int NP, int OM>
__global__ __launch_bounds__(256, 1) void wgrad_split_kernel(WgradLaunch a, int tilesY, int tilesX, int numTiles) {
  static_assert(WM * WN * WK == 4, "4 waves per workgroup");
  static_assert(OM == 0 || NP == 1, "16-bit storage: one MFMA per product");
  constexpr bool X16 = OM == 1 || OM == 2, D16 = OM >= 1;       // operand storage, see wg_mfma above
  constexpr int CI = 16 * MI * WM, CO = 16 * NI * WN;
  constexpr int PA = wg_pitch(CI), PD = wg_pitch(CO);
  // 16-byte units per pixel: fp32 activations 4 channels each, 16-bit activations 8; dz pre-split: hi and lo halves of
  // 8 channels each, plain 16-bit dz: 8 channels each
  constexpr int QA = X16 ? CI / 8 : CI / 4, QD = D16 ? CO / 8 : CO / 4;
  constexpr int XA = (kWgAPix * QA + 255) / 256, XD = (kWgDPix * QD + 255) / 256;
  constexpr int ABYTES = kWgAPix * PA, DBYTES = kWgDPix * PD;
  constexpr int REDBYTES = 4 * MI * NI * 256 * 4;               // cross-wave reduction scratch
  constexpr int LDSBYTES = ABYTES + DBYTES > REDBYTES ? ABYTES + DBYTES : REDBYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDSBYTES];
  unsigned char* as_ = smem;
  unsigned char* ds_ = smem + ABYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int q = lr >> 2, p4 = lr & 3;
  const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
  const int coTiles = a.cout_pad / CO;
  const int ciT = blockIdx.x / coTiles, coT = blockIdx.x - ciT * coTiles;
  const int ci0 = ciT * CI, co0 = coT * CO;

  f32x4 acc[9][MI][NI];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[t][mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 xa[XA], xd[XD];
#define WG_LOAD(TILE)                                                                               \
  {                                                                                                 \
    int t_ = (TILE);                                                                                \
    const int tx_ = t_ % tilesX;                                                                    \
    t_ /= tilesX;                                                                                   \
    const int ty_ = t_ % tilesY;                                                                    \
    const int n_ = t_ / tilesY;                                                                     \
    const int y0_ = ty_ * kWgTR, x0_ = tx_ * kWgTC;                                                 \
    const float* ximg_ = a.x + (size_t)n_ * a.H * a.W * a.ldx;                                      \
    const float* dimg_ = a.dz + (size_t)n_ * a.H * a.W * a.lddz;                                    \
    _Pragma("unroll") for (int k_ = 0; k_ < XA; ++k_) {                                             \
      const int u_ = min(tid + k_ * 256, kWgAPix * QA - 1);                                         \
      const int pix_ = u_ / QA, qq_ = u_ - pix_ * QA;                                               \
      const int tr_ = pix_ / kWgTCP, tc_ = pix_ - tr_ * kWgTCP;                                     \
      int iy_ = y0_ - 1 + tr_, ix_ = x0_ - 1 + tc_;                                                 \
      iy_ = iy_ < 0 ? -iy_ : iy_;                                                                   \
      iy_ = iy_ >= a.H ? 2 * a.H - 2 - iy_ : iy_;                                                   \
      ix_ = ix_ < 0 ? -ix_ : ix_;                                                                   \
      ix_ = ix_ >= a.W ? 2 * a.W - 2 - ix_ : ix_;                                                   \
      iy_ = min(max(iy_, 0), a.H - 1);                                                              \
      ix_ = min(max(ix_, 0), a.W - 1);                                                              \
      if (X16) {                                                                                    \
        const bool ok_ = ci0 + 8 * qq_ < a.cin_p;                                                   \
        const unsigned short* s_ = reinterpret_cast<const unsigned short*>(a.x) +                   \
                                   (((size_t)n_ * a.H + iy_) * a.W + ix_) * a.ldx + ci0 + 8 * qq_;  \
        xa[k_] = *reinterpret_cast<const f32x4*>(ok_ ? reinterpret_cast<const float*>(s_) : kZeroPage); \
      } else {                                                                                      \
        const bool ok_ = ci0 + 4 * qq_ < a.cin_p; /* masked-out units load from kZeroPage (common.h) */ \
        xa[k_] = *reinterpret_cast<const f32x4*>(ok_ ? ximg_ + ((size_t)iy_ * a.W + ix_) * a.ldx + ci0 + 4 * qq_ : kZeroPage); \
      }                                                                                             \
    }                                                                                               \
    _Pragma("unroll") for (int k_ = 0; k_ < XD; ++k_) {                                             \
      const int u_ = tid + k_ * 256;                                                                \
      const int pix_ = u_ / QD, qq_ = u_ - pix_ * QD;                                               \
      const int r_ = pix_ / kWgTC, c_ = pix_ - r_ * kWgTC;                                          \
      const int y_ = y0_ + r_, x_ = x0_ + c_;                                                       \
      if (D16) { /* plain 16-bit NHWC dz: unit = 8 channels */                                      \
        const int ch_ = co0 + 8 * qq_;                                                              \
        const bool ok_ = pix_ < kWgDPix && y_ < a.H && x_ < a.W && ch_ < a.cout_p;                  \
        const unsigned short* s_ = reinterpret_cast<const unsigned short*>(a.dz) +                  \
                                   (((size_t)n_ * a.H + y_) * a.W + x_) * a.lddz + ch_;             \
        xd[k_] = *reinterpret_cast<const f32x4*>(ok_ ? reinterpret_cast<const float*>(s_) : kZeroPage); \
      } else {                                                                                      \
      /* dz is pre-split, per 32-channel chunk [hi rc | lo rc] bf16: unit = 8 channels, 16 bytes */ \
      const int half_ = qq_ / (CO / 8), ch_ = co0 + 8 * (qq_ - half_ * (CO / 8));                   \
      const int rc_ = min(32, a.cout_p - (ch_ & ~31));                                              \
      const bool ok_ = pix_ < kWgDPix && y_ < a.H && x_ < a.W && ch_ < a.cout_p && (NP == 3 || half_ == 0); \
      const unsigned short* s_ = reinterpret_cast<const unsigned short*>(dimg_ + ((size_t)y_ * a.W + x_) * a.lddz) + \
                                 ((ch_ >> 5) * 64 + half_ * rc_ + (ch_ & 31));                      \
      xd[k_] = *reinterpret_cast<const f32x4*>(ok_ ? reinterpret_cast<const float*>(s_) : kZeroPage); \
      }                                                                                             \
    }                                                                                               \
  }
#define WG_SPLIT_STORE(V, DST, CCH)                                                                 \
  {                                                                                                 \
    if (OM != 0) { /* 16-bit modes: one rounding to the element type, no lo part */                 \
      *reinterpret_cast<bf16x4*>(DST) = wg_round4<OM>(V);                                           \
    } else {                                                                                        \
      bf16x4 hi_, lo_;                                                                              \
      _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                            \
        hi_[e_] = (__bf16)(V)[e_];                                                                  \
        lo_[e_] = (__bf16)((V)[e_] - (float)hi_[e_]);                                               \
      }                                                                                             \
      *reinterpret_cast<bf16x4*>(DST) = hi_;                                                        \
      if (NP == 3) *reinterpret_cast<bf16x4*>((DST) + 2 * (CCH)) = lo_;                             \
    }                                                                                               \
  }
#define WG_STORE()                                                                                  \
  {                                                                                                 \
    _Pragma("unroll") for (int k_ = 0; k_ < XA; ++k_) {                                             \
      const int u_ = tid + k_ * 256;                                                                \
      const int pix_ = u_ / QA, qq_ = u_ - pix_ * QA;                                               \
      if (pix_ < kWgAPix) {                                                                         \
        if (X16) { /* 8 channels of the hi plane: plain 16-byte copy */                             \
          *reinterpret_cast<f32x4*>(as_ + pix_ * PA + ((pix_ >> 3) & 1) * 32 + qq_ * 16) = xa[k_];  \
        } else {                                                                                    \
          unsigned char* d_ = as_ + pix_ * PA + ((pix_ >> 3) & 1) * 32 + qq_ * 8;                   \
          WG_SPLIT_STORE(xa[k_], d_, CI)                                                            \
        }                                                                                           \
      }                                                                                             \
    }                                                                                               \
    _Pragma("unroll") for (int k_ = 0; k_ < XD; ++k_) {                                             \
      const int u_ = tid + k_ * 256;                                                                \
      const int pix_ = u_ / QD, qq_ = u_ - pix_ * QD;                                               \
      if (pix_ < kWgDPix) { /* row image [hi CO | lo CO]: plain 16-byte copy of the pre-split dz */ \
        const int half_ = D16 ? 0 : qq_ / (CO / 8), c8_ = qq_ - half_ * (CO / 8);                   \
        unsigned char* d_ = ds_ + pix_ * PD + ((pix_ >> 3) & 1) * 32 + half_ * 2 * CO + c8_ * 16;   \
        *reinterpret_cast<f32x4*>(d_) = xd[k_];                                                     \
      }                                                                                             \
    }                                                                                               \
  }

  int tile = blockIdx.y;
  if (tile < numTiles) {
    WG_LOAD(tile)
    WG_STORE()
  }
  __syncthreads();
  // per-lane row / column constants of the transposed reads
  const int rowk = 8 * g + q;                          // K index (pixel column inside the tile row), first half
  const int dcol = (wn * NI) * 32 + 8 * p4 + 32 * (g & 1);  // dz: rows r*32 + rowk (+4): bit 3 == g & 1
  const int acol = (wm * MI) * 32 + 8 * p4;

  for (; tile < numTiles; tile += gridDim.y) {
    const int next = tile + gridDim.y;
    if (next < numTiles) WG_LOAD(next)
#pragma unroll
    for (int rr = 0; rr < kWgTR / WK; ++rr) {
      const int r = wk + rr * WK;
      bf16x8 bh[NI], bl[NI];
      const unsigned char* d0 = ds_ + (r * kWgTC + rowk) * PD + dcol;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        bh[ni] = tr_read8(d0 + ni * 32, d0 + ni * 32 + 4 * PD);
        if (NP == 3) bl[ni] = tr_read8(d0 + ni * 32 + 2 * CO, d0 + ni * 32 + 2 * CO + 4 * PD);
      }
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int row0 = (r + kh) * kWgTCP + rowk + kw, row1 = row0 + 4;
          const unsigned char* a0 = as_ + row0 * PA + ((row0 >> 3) & 1) * 32 + acol;
          const unsigned char* a1 = as_ + row1 * PA + ((row1 >> 3) & 1) * 32 + acol;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const bf16x8 ah = tr_read8(a0 + mi * 32, a1 + mi * 32);
            bf16x8 al = ah;
            if (NP == 3) al = tr_read8(a0 + mi * 32 + 2 * CI, a1 + mi * 32 + 2 * CI);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              f32x4 c = acc[kh * 3 + kw][mi][ni];
              if (NP == 3) {
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[ni], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[ni], c, 0, 0, 0);
              }
              c = wg_mfma<OM>(ah, bh[ni], c);
              acc[kh * 3 + kw][mi][ni] = c;
            }
          }
        }
      }
    }
    __syncthreads();  // every wave is done reading this tile
    if (next < numTiles) WG_STORE()
    __syncthreads();
  }
#undef WG_LOAD
#undef WG_STORE
#undef WG_SPLIT_STORE

  // ---- write the partial slab (after combining the WK waves that share a (ci,co) tile) -------
  float* out = a.partial + (size_t)blockIdx.y * 9 * a.cin_pad * a.cout_pad;
  float* red = reinterpret_cast<float*>(smem);  // [wave][mi][ni][256]
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    if (WK > 1) {
      __syncthreads();
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          *reinterpret_cast<f32x4*>(red + ((wave * MI + mi) * NI + ni) * 256 + lane * 4) = acc[t][mi][ni];
      __syncthreads();
    }
    if (wk == 0) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          f32x4 v = acc[t][mi][ni];
          if (WK > 1) {
            v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < WK; ++k) v += *reinterpret_cast<const f32x4*>(red + (((wave + k) * MI + mi) * NI + ni) * 256 + lane * 4);
          }
          const int co = co0 + (wn * NI + ni) * 16 + lr;
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            const int ci = ci0 + (wm * MI + mi) * 16 + g * 4 + r4;
            out[((size_t)t * a.cin_pad + ci) * a.cout_pad + co] = v[r4];
          }
        }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Wave-specialised variant for the 64x64 (ci,co) tile — the configuration that carries ~80 % of the
// weight-gradient time.  PMC counters of the kernel above: 3.3 VALU per MFMA with ONE wave per SIMD,
// i.e. the staging work (address arithmetic, fp32 -> hi/lo split, LDS writes) sits in the same
// instruction stream as the MFMAs and the matrix pipe idles about half the time.  Here waves 0-3
// ("consumers") only issue transposed LDS reads + MFMAs and waves 4-7 ("producers", one per SIMD
// next to a consumer) do all the staging into a double-buffered LDS tile, so the VALU and the MFMA
// pipes of every SIMD run concurrently.  Tile = 2 rows x 32 columns (K = 64 pixels), one barrier
// per tile; producers run one tile ahead in LDS and two tiles ahead in registers.
// ---------------------------------------------------------------------------------------
// pixel tile of the wave-specialised kernel: TR rows x 32 columns.  2 rows with 64 input channels per
// workgroup (128 KB of LDS); 4 rows with 32 (the thin 256x256 layers: twice the MFMAs per barrier, 127-157 KB) and,
// round 3, 4 rows when both operands are 16-bit tensors (rows of one 16-bit plane: 128 KB again; twice the MFMAs
// per barrier, 1.5 x instead of 2 x halo rows of the activation operand)
static bool wgrad_s16(int om) { return om == 1 || om == 2; }
static int wgrad_ws_tr(int CI, int om) { return sched::wg_ws_tr(CI, wgrad_s16(om)); }

// FIN (split16 only): the activation operand is the producing convolution's PRE-activation tensor and the producers apply
// its BatchNorm + ReLU (WgradLaunch::in_scale / in_shift) in front of the bf16 split: relu(fma(z, scale, shift)), exactly
// bn_relu_fwd_kernel's arithmetic, so the activated tensor need not exist in HBM.  A lane's units all hold the same
// channel quad (256 threads step over whole pixels), so the eight constants are loaded once per thread.
template <int NP, int NI, int CI_, int TR_, int OM, bool FIN = false>
__global__ __launch_bounds__(512, 2) void wgrad_split_ws_kernel(WgradLaunch a, int tilesY, int tilesX, int numTiles) {
  static_assert(OM == 0 || NP == 1, "16-bit storage: one MFMA per product");
  static_assert(!FIN || (OM == 0 && NP >= 2), "fused input BatchNorm + ReLU: split16 only");
  // NP == 2 (round 5): TWO MFMAs per product on fp16 operands — the activation as ONE fp16 value (11 significant bits: 2^-12
  // per element, an error that stays in this layer's weight gradient), dz as an fp16 (hi, lo) pair: a.dz_hi + a.dz_lo — a
  // third fewer MFMAs and three instead of four transposed fragment reads per pair, the two things that bound this kernel
  // (profiles/r04/wgrad_read_pipeline.txt).  dz still arrives as bf16 pair records (the data gradient keeps its arithmetic);
  // the producers scale both halves of every (hi, lo) bf16 pair by the power of two that puts the layer's largest |dz|
  // (WgradLaunch::dz_absmax, written by the BatchNorm backward) into [2^14, 2^15) and convert them to fp16 — exactly, 8
  // significant bits each (wg_dz_pair_to_f16); the slabs carry that factor and the reduction removes it.
  constexpr bool X16 = OM == 1 || OM == 2, D16 = OM >= 1;  // operand storage, see wg_mfma above
  // consumer wave = 16 ci x CO co (MI = 1, NI = CO/16 = 2, 3 or 4): an A fragment (re-read for every tap)
  // feeds 3*NI MFMAs; at NI = 4, 52 instead of 80 transposed LDS reads per 108 MFMAs of a 32x32 arrangement.
  // CI_ = 64: the four consumer waves are stacked along ci, each accumulates all nine taps.  CI_ = 32 (layers
  // with <= 32 input channels): two waves along ci x two tap sets (taps 0-4 / 5-8), so all four SIMDs still
  // multiply and a wave holds 5 instead of 9 tap accumulators.
  constexpr int CI = CI_, CO = 16 * NI, MI = 1;
  constexpr bool TSPLIT = CI == 32;
  static_assert(CI == 64 || CI == 32, "64 or 32 input channels per workgroup");
  constexpr int kWsTR = TR_, kWsAPix = (kWsTR + 2) * kWgTCP, kWsDPix = kWsTR * kWgTC;
  constexpr int PA = X16 ? wg_pitch16(CI) : wg_pitch(CI), PD = D16 ? wg_pitch16(CO) : wg_pitch(CO);
  constexpr int QA = X16 ? CI / 8 : CI / 4, QD = D16 ? CO / 8 : CO / 4;  // 16-byte units per pixel
#ifdef MIMO_WGRAD_ABLATE
  // timing-only builds (results are wrong): 1 = the producers stage only the first half of the activation halo tile (what
  // a row ring that shares halo rows between vertically adjacent tiles would stage per tile); 2 = none of it; 4 = no dz
  constexpr int XA = (MIMO_WGRAD_ABLATE & 2) ? 1 : (MIMO_WGRAD_ABLATE & 1) ? (kWsAPix * QA / 2 + 255) / 256 : (kWsAPix * QA + 255) / 256;
  constexpr int XD = (MIMO_WGRAD_ABLATE & 4) ? 1 : (kWsDPix * QD + 255) / 256;
#else
  constexpr int XA = (kWsAPix * QA + 255) / 256;  // per producer thread
  constexpr int XD = NP == 2 ? 2 * ((kWsDPix * (CO / 8) + 255) / 256) : (kWsDPix * QD + 255) / 256;
#endif
  constexpr int ABYTES = kWsAPix * PA, DBYTES = kWsDPix * PD, BUFBYTES = ABYTES + DBYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BUFBYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int coTiles = a.cout_pad / CO;
  // 1-D grid of wtiles * splits workgroups.  XCD-aware order (xcd_virtual_index, common.h): virtual index =
  // split * wtiles + weight tile, so the (ci, co) weight tiles of one pixel split — which all read the same
  // activation / dz tiles — occupy as few XCDs as possible and fetch those tiles from HBM once per XCD
  // (before: every XCD held one co tile of every split, i.e. the activation operand was read 8 x).
  const int wtiles = (a.cin_pad / CI) * coTiles;
  const int v_ = xcd_virtual_index((int)blockIdx.x, wtiles * a.splits);
  const int bsplit = v_ / wtiles, bwt = v_ - bsplit * wtiles;
  const int ciT = bwt / coTiles, coT = bwt - ciT * coTiles;
  const int ci0 = ciT * CI, co0 = coT * CO;
  const int ntiles_mine = bsplit < numTiles ? (numTiles - 1 - bsplit) / a.splits + 1 : 0;

  if (wave >= 4) {
    // =========================== producers ===========================
    const int ptid = tid - 256;
    f32x4 xa0[XA], xd0[XD], xa1[XA], xd1[XD];  // two tiles in flight: global latency > one tile of MFMAs
    // per-unit constants, hoisted out of the tile loop: halo-tile coordinates, channel, LDS destination
    int a_tr[XA], a_tc[XA], a_ch[XA], a_dst[XA], d_r[XD], d_c[XD], d_ch[XD], d_dst[XD];
#pragma unroll
    for (int k = 0; k < XA; ++k) {
      const int u = ptid + k * 256;
      // units past the tile (XA rounds up) repeat the thread's OWN previous unit: same lane, same channel quad
      static_assert(256 % QA == 0 && kWsAPix * QA >= 256, "a thread's units share one channel quad");
      const int uc = u < kWsAPix * QA ? u : u - 256;
      const int pix = uc / QA, qq = uc - pix * QA;
      a_tr[k] = pix / kWgTCP - 1;
      a_tc[k] = pix % kWgTCP - 1;
      constexpr int kChU = X16 ? 8 : 4;  // channels per unit; its hi-plane bytes = 2 * kChU
      a_ch[k] = ci0 + kChU * qq < a.cin_p ? ci0 + kChU * qq : -1;
      // units past the tile repeat an earlier unit — same source (uc), same destination, same bytes: every
      // store below is unconditional.  A store skipped by a branch leaves its load un-waited on that path and hipcc then
      // drains the whole queue (vmcnt(0)) before the registers are reused, which cut the prefetch from two tiles to one
      // (found in the ISA in round 3, as in conv_wide.hip).
      a_dst[k] = pix * PA + ((pix >> 3) & 1) * 32 + qq * 2 * kChU;
    }
#pragma unroll
    for (int k = 0; k < XD; ++k) {
      if constexpr (NP == 2) {
        // pair units: unit 2 kp = the hi plane, 2 kp + 1 = the lo plane of the SAME 8 channels of one pixel (the thread
        // re-splits the pair), kp over (pixel, 8-channel group)
        const int kp = k >> 1, is_lo = k & 1;
        const int u = min(ptid + kp * 256, kWsDPix * (CO / 8) - 1);
        const int pix = u / (CO / 8), c8 = u - pix * (CO / 8);
        d_r[k] = pix / kWgTC;
        d_c[k] = pix % kWgTC;
        const int ch = co0 + 8 * c8, rc = min(32, a.cout_p - (ch & ~31));
        d_ch[k] = ch < a.cout_p ? (ch >> 5) * 64 + is_lo * rc + (ch & 31) : -1;
        d_dst[k] = ABYTES + pix * PD + ((pix >> 3) & 1) * 32 + is_lo * 2 * CO + c8 * 16;
      } else {
      const int u = min(ptid + k * 256, kWsDPix * QD - 1);  // units past the tile repeat its last unit (see a_dst)
      const int pix = u / QD, qq = u - pix * QD;
      const int half = D16 ? 0 : qq / (CO / 8), c8 = qq - half * (CO / 8);  // pre-split dz: 8 channels of the hi or lo plane
      d_r[k] = pix / kWgTC;
      d_c[k] = pix % kWgTC;
      const int ch = co0 + 8 * c8, rc = min(32, a.cout_p - (ch & ~31));  // chunk record [hi rc | lo rc]
      d_ch[k] = (pix < kWsDPix && ch < a.cout_p && (NP == 3 || half == 0))
                    ? (D16 ? ch : (ch >> 5) * 64 + half * rc + (ch & 31))  // plain NHWC / pair records, in 16-bit units
                    : -1;
      d_dst[k] = ABYTES + pix * PD + ((pix >> 3) & 1) * 32 + half * 2 * CO + c8 * 16;
      }
    }
    // NP == 2: the power of two that scales this layer's dz into fp16 range (see the kernel comment)
    const float dz_mul = NP == 2 ? wg_dz_scale(__float_as_uint(wg_dz_absmax(a.dz_absmax, a.dz_absmax_n)), false) : 1.f;
    (void)dz_mul;
    f32x4 in_sc = f32x4{0.f, 0.f, 0.f, 0.f}, in_sh = f32x4{0.f, 0.f, 0.f, 0.f};
    if (FIN && a_ch[0] >= 0) {  // (channels past cin_p are loaded from the zero page and must stay zero: relu(0 * 0 + 0))
      in_sc = *reinterpret_cast<const f32x4*>(a.in_scale + a_ch[0]);
      in_sh = *reinterpret_cast<const f32x4*>(a.in_shift + a_ch[0]);
    }
    const int H2 = 2 * a.H - 2, W2 = 2 * a.W - 2;
#define WS_LOAD(XA_, XD_, TILE)                                                                     \
  {                                                                                                 \
    int t_ = (TILE);                                                                                \
    const int tx_ = t_ % tilesX;                                                                    \
    t_ /= tilesX;                                                                                   \
    const int ty_ = t_ % tilesY;                                                                    \
    const int n_ = t_ / tilesY;                                                                     \
    const int y0_ = ty_ * kWsTR, x0_ = tx_ * kWgTC;                                                 \
    const float* ximg_ = a.x + (size_t)n_ * a.H * a.W * a.ldx;                                      \
    const float* dimg_ = a.dz + (size_t)n_ * a.H * a.W * a.lddz;                                    \
    _Pragma("unroll") for (int k_ = 0; k_ < XA; ++k_) {                                             \
      int iy_ = y0_ + a_tr[k_], ix_ = x0_ + a_tc[k_];                                               \
      iy_ = max(iy_, -iy_);           /* reflect at 0 */                                            \
      iy_ = max(min(iy_, H2 - iy_), 0); /* reflect at H-1; clamp the tile overhang */               \
      ix_ = max(ix_, -ix_);                                                                         \
      ix_ = max(min(ix_, W2 - ix_), 0);                                                             \
      const bool ok_ = a_ch[k_] >= 0;                                                               \
      if (X16) {                                                                                    \
        const unsigned short* s_ = reinterpret_cast<const unsigned short*>(a.x) +                   \
                                   (((size_t)n_ * a.H + iy_) * a.W + ix_) * a.ldx + a_ch[k_];       \
        XA_[k_] = *reinterpret_cast<const f32x4*>(ok_ ? reinterpret_cast<const float*>(s_) : kZeroPage); \
      } else {                                                                                      \
        XA_[k_] = *reinterpret_cast<const f32x4*>(ok_ ? ximg_ + (iy_ * a.W + ix_) * a.ldx + a_ch[k_] : kZeroPage); \
      }                                                                                             \
    }                                                                                               \
    _Pragma("unroll") for (int k_ = 0; k_ < XD; ++k_) {                                             \
      const int y_ = y0_ + d_r[k_], x_ = x0_ + d_c[k_];                                             \
      const bool ok_ = d_ch[k_] >= 0 && y_ < a.H && x_ < a.W;                                       \
      const unsigned short* s_ = D16 ? reinterpret_cast<const unsigned short*>(a.dz) +              \
                                           (((size_t)n_ * a.H + y_) * a.W + x_) * a.lddz + d_ch[k_] \
                                     : reinterpret_cast<const unsigned short*>(dimg_ + (y_ * a.W + x_) * a.lddz) + d_ch[k_]; \
      XD_[k_] = *reinterpret_cast<const f32x4*>(ok_ ? reinterpret_cast<const float*>(s_) : kZeroPage); \
    }                                                                                               \
  }
#define WS_SPLIT_STORE(V, DST, CCH)                                                                 \
  {                                                                                                 \
    if (X16) { /* 8 channels of the hi plane: plain 16-byte copy */                                 \
      *reinterpret_cast<f32x4*>(DST) = (V);                                                         \
    } else if (OM != 0) { /* fp32 image operand in a 16-bit mode: one rounding, no lo part */       \
      *reinterpret_cast<bf16x4*>(DST) = wg_round4<OM>(V);                                           \
    } else if (NP == 2) { /* one fp16 value per activation */                                      \
      f16x4_w h_;                                                                                   \
      _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_)                                              \
        h_[e_] = (_Float16)(FIN ? fmaxf(fmaf((V)[e_], in_sc[e_], in_sh[e_]), 0.f) : (V)[e_]);       \
      *reinterpret_cast<f16x4_w*>(DST) = h_;                                                        \
    } else {                                                                                        \
      bf16x4 hi_, lo_;                                                                              \
      _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                            \
        const float v_ = FIN ? fmaxf(fmaf((V)[e_], in_sc[e_], in_sh[e_]), 0.f) : (V)[e_];           \
        hi_[e_] = (__bf16)v_;                                                                       \
        lo_[e_] = (__bf16)(v_ - (float)hi_[e_]);                                                    \
      }                                                                                             \
      *reinterpret_cast<bf16x4*>(DST) = hi_;                                                        \
      if (NP == 3) *reinterpret_cast<bf16x4*>((DST) + 2 * (CCH)) = lo_;                             \
    }                                                                                               \
  }
#define WS_STORE(XA_, XD_, BUF)                                                                      \
  {                                                                                                 \
    unsigned char* base_ = smem + (BUF) * BUFBYTES;                                                 \
    _Pragma("unroll") for (int k_ = 0; k_ < XA; ++k_) {                                             \
      unsigned char* d_ = base_ + a_dst[k_];                                                        \
      WS_SPLIT_STORE(XA_[k_], d_, CI)                                                               \
    }                                                                                               \
    if (NP == 2) { /* bf16 (hi, lo) records -> scaled fp16 (hi, lo) */                              \
      _Pragma("unroll") for (int k_ = 0; k_ + 1 < XD; k_ += 2) {                                    \
        f32x4 oh_, ol_;                                                                             \
        wg_dz_pair_to_f16(XD_[k_], XD_[k_ + 1], dz_mul, &oh_, &ol_);                                \
        *reinterpret_cast<f32x4*>(base_ + d_dst[k_]) = oh_;                                         \
        *reinterpret_cast<f32x4*>(base_ + d_dst[k_ + 1]) = ol_;                                     \
      }                                                                                             \
    } else _Pragma("unroll") for (int k_ = 0; k_ < XD; ++k_) {                                      \
      *reinterpret_cast<f32x4*>(base_ + d_dst[k_]) = XD_[k_]; /* plain copy */                      \
    }                                                                                               \
  }
    // register set s (0/1) carries tile j with j&1 == s; loads are issued two tiles (= two barriers) ahead
    // Straight-line, unconditional: past the end the loads re-read the last tile and the stores go to a buffer nobody
    // reads any more (a conditional load or store makes hipcc wait vmcnt(0) where the paths join).  The loop always
    // runs an even number of phases; the consumers add a barrier when their tile count is odd.
    const int T0 = bsplit, TS = a.splits, last = max(ntiles_mine - 1, 0);
#define WS_TILE(J) (T0 + min((J), last) * TS)
    WS_LOAD(xa0, xd0, WS_TILE(0))
    WS_LOAD(xa1, xd1, WS_TILE(1))
    WS_STORE(xa0, xd0, 0)
    WS_LOAD(xa0, xd0, WS_TILE(2))
    __syncthreads();  // tile 0 is in LDS
    for (int i = 0; i < ntiles_mine; i += 2) {
      // consumers multiply tile i from buffer i&1; the other buffer was released by the previous barrier
      WS_STORE(xa1, xd1, 1)
      WS_LOAD(xa1, xd1, WS_TILE(i + 3))
      __syncthreads();
      WS_STORE(xa0, xd0, 0)
      WS_LOAD(xa0, xd0, WS_TILE(i + 4))
      __syncthreads();
    }
#undef WS_TILE
#undef WS_LOAD
#undef WS_STORE
#undef WS_SPLIT_STORE
    return;
  }

  // =========================== consumers ===========================
  const int lr = lane & 15, g = lane >> 4;
  const int q = lr >> 2, p4 = lr & 3;
  const int wn = 0, wm = TSPLIT ? (wave & 1) : wave;
  const int tset = TSPLIT ? (wave >> 1) : 0;  // wave-uniform
  static_assert(NI * 16 == CO, "one co tile per workgroup");
  constexpr int NTMAX = TSPLIT ? 5 : 9;
  f32x4 acc[NTMAX][MI][NI];
#pragma unroll
  for (int t = 0; t < NTMAX; ++t)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[t][mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int rowk = 8 * g + q;
  const int dcol = (wn * NI) * 32 + 8 * p4 + 32 * (g & 1);
  const int acol = (wm * MI) * 32 + 8 * p4;

  __syncthreads();  // tile 0 is in LDS
  for (int i = 0; i < ntiles_mine; ++i) {
    const unsigned char* as_ = smem + (i & 1) * BUFBYTES;
    const unsigned char* ds_ = as_ + ABYTES;
    {
      // The whole tile is one unrolled sequence of TR x taps steps: the activation fragments are read kDepth taps ahead
      // and the pipeline runs across the tile's rows; the dz fragments of row r + 1 are read under the first tap of row
      // r.  One MFMA per product (16-bit storage, bf16 mode): a tap is NI MFMAs = 16 * NI cycles of matrix-pipe time,
      // far below the ~130-cycle latency of a transposed LDS read -> two taps ahead (round 3: with one tap ahead that
      // path waited for LDS at every tap, 598 -> 789 TFLOP/s on the class).  Three MFMAs per product: one tap ahead.
      constexpr int kDepth = NP == 3 ? 1 : NP == 2 ? MIMO_WGRAD_NP2_DEPTH : 2;
      bf16x8 bh[2][NI], bl[NP >= 2 ? 2 : 1][NI], ah[kDepth + 1][MI], al[NP == 3 ? kDepth + 1 : 1][MI];
      // the fragment addresses of all TR x taps steps are loop-invariant; hoisted out of the tile loop they would take
      // ~45 registers (spills) — an opaque copy of the lane's row index per tile keeps them recomputed in place
      int rowk_ = rowk;
      asm volatile("" : "+v"(rowk_));
#define WS_READ_B(SLOT, R)                                                           \
  {                                                                                  \
    const unsigned char* d0_ = ds_ + ((R) * kWgTC + rowk_) * PD + dcol;              \
    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                              \
      bh[SLOT][ni] = tr_read8(d0_ + ni * 32, d0_ + ni * 32 + 4 * PD);                \
      if (NP >= 2) bl[SLOT][ni] = tr_read8(d0_ + ni * 32 + 2 * CO, d0_ + ni * 32 + 2 * CO + 4 * PD); \
    }                                                                                \
  }
#define WS_READ_A1(SLOT, R, TAP)                                                     \
  {                                                                                  \
    const int row0_ = ((R) + (TAP) / 3) * kWgTCP + rowk_ + (TAP) % 3, row1_ = row0_ + 4; \
    const unsigned char* a0_ = as_ + row0_ * PA + ((row0_ >> 3) & 1) * 32 + acol;    \
    const unsigned char* a1_ = as_ + row1_ * PA + ((row1_ >> 3) & 1) * 32 + acol;    \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                              \
      ah[SLOT][mi] = tr_read8(a0_ + mi * 32, a1_ + mi * 32);                         \
      if (NP == 3) al[SLOT][mi] = tr_read8(a0_ + mi * 32 + 2 * CI, a1_ + mi * 32 + 2 * CI); \
    }                                                                                \
  }
#define WS_TILE1(T0, NT)                                                             \
  {                                                                                  \
    constexpr int kSteps = kWsTR * (NT);                                             \
    constexpr int kRB = (NP >= 2 ? 4 : 2) * NI, kRA = (NP == 3 ? 4 : 2) * MI;        \
    WS_READ_B(0, 0)                                                                  \
    _Pragma("unroll") for (int s_ = 0; s_ < kDepth; ++s_)                            \
      WS_READ_A1(s_ % (kDepth + 1), s_ / (NT), (T0) + s_ % (NT))                     \
    /* Without this pin the FIRST in-loop group (the reads of step kDepth) takes the prologue's reads and every later \
       group moves one step down: in the ISA the fragments of a tap are read right in front of its MFMAs.  With it   \
       they are read one tap ahead, as the source says — and the kernel is 3.5 % slower (see the switch's comment). */ \
    if (MIMO_WGRAD_PIN_PROLOGUE) __builtin_amdgcn_sched_group_barrier(0x100, kRB + kDepth * kRA, 0); \
    _Pragma("unroll") for (int s_ = 0; s_ < kSteps; ++s_) {                          \
      const int r_ = s_ / (NT), tt = s_ % (NT);                                      \
      const bool rb_ = tt == 0 && r_ + 1 < kWsTR, ra_ = s_ + kDepth < kSteps;        \
      if (rb_) WS_READ_B((r_ + 1) & 1, r_ + 1)                                       \
      if (ra_) WS_READ_A1((s_ + kDepth) % (kDepth + 1), (s_ + kDepth) / (NT), (T0) + (s_ + kDepth) % (NT)) \
      _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                              \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                          \
          f32x4 c = acc[tt][mi][ni];                                                 \
          if (NP == 3) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[s_ % (kDepth + 1)][mi], bh[r_ & 1][ni], c, 0, 0, 0); \
          if (NP == 3) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[s_ % (kDepth + 1)][mi], bl[r_ & 1][ni], c, 0, 0, 0); \
          if (NP == 2) c = wg_mfma<2>(ah[s_ % (kDepth + 1)][mi], bl[r_ & 1][ni], c);  /* fp16: a . dz_lo */ \
          acc[tt][mi][ni] = wg_mfma<(NP == 2 ? 2 : OM)>(ah[s_ % (kDepth + 1)][mi], bh[r_ & 1][ni], c); \
        }                                                                            \
      if (rb_ && ra_) {                                                              \
        __builtin_amdgcn_sched_group_barrier(0x100, kRB + kRA, 0);                   \
      } else if (rb_) {                                                              \
        __builtin_amdgcn_sched_group_barrier(0x100, kRB, 0);                         \
      } else if (ra_) {                                                              \
        __builtin_amdgcn_sched_group_barrier(0x100, kRA, 0);                         \
      }                                                                              \
      __builtin_amdgcn_sched_group_barrier(0x008, NP * MI * NI, 0);                  \
    }                                                                                \
  }
      if (!TSPLIT) {
        WS_TILE1(0, 9)
      } else if (tset == 0) {
        WS_TILE1(0, 5)
      } else {
        WS_TILE1(5, 4)
      }
#undef WS_TILE1
#undef WS_READ_A1
#undef WS_READ_B
    }
    __syncthreads();
  }
  if (ntiles_mine & 1) __syncthreads();  // the producers' loop runs an even number of phases
  float* out = a.partial + (size_t)bsplit * 9 * a.cin_pad * a.cout_pad;
  const int t0 = tset ? 5 : 0, nt = TSPLIT ? (tset ? 4 : 5) : 9;
#pragma unroll
  for (int tt = 0; tt < NTMAX; ++tt)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int co = co0 + (wn * NI + ni) * 16 + lr;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int ci = ci0 + (wm * MI + mi) * 16 + g * 4 + r4;
          if (tt < nt) out[((size_t)(t0 + tt) * a.cin_pad + ci) * a.cout_pad + co] = acc[tt][mi][ni][r4];
        }
      }
}

// host-side choices (channel tiles, split counts): tile_sched.h, host-testable
static bool wgrad_ws_enabled() {
  static const bool on = !(getenv("MIMO_WGRAD_WS") && atoi(getenv("MIMO_WGRAD_WS")) == 0);
  return on;
}
void wgrad_split_tiles(int cin_p, int cout_p, int* CI, int* CO) { sched::wg_tiles(cin_p, cout_p, wgrad_ws_enabled(), CI, CO); }
static bool wgrad_use_ws(int CI, int CO) { return sched::wg_use_ws(CI, CO, wgrad_ws_enabled()); }
int wgrad_split_pick_splits(int N, int H, int W, int cin_pad, int cout_pad, int CI, int CO, int store, int cus) {
  static const int mode = [] { const char* e = getenv("MIMO_WGRAD_SPLIT_MODE"); return e ? atoi(e) : 1; }();
  return sched::wg_pick_splits(N, H, W, cin_pad, cout_pad, CI, CO, wgrad_ws_enabled(), mode, wgrad_s16(store),
                               cus >= 8 && cus <= 256 ? cus : 256);
}

// 1 when this geometry runs on the kernel that has the two-MFMA (fp16) instances: WgradLaunch::np = 2
int wgrad_split_has_np2(int cin_p, int cout_p) {
  const bool on = !(getenv("MIMO_WGRAD_NP") && atoi(getenv("MIMO_WGRAD_NP")) == 3);  // =3: three bf16 MFMAs (rounds 1-4); read per plan / call
  int CI, CO;
  wgrad_split_tiles(cin_p, cout_p, &CI, &CO);
  return on && wgrad_use_ws(CI, CO) ? 1 : 0;
}

int wgrad_split_fuses_input(int cin_p, int cout_p, int store, int np) {
  int CI, CO;
  wgrad_split_tiles(cin_p, cout_p, &CI, &CO);
  return wgrad_use_ws(CI, CO) && store == 0 && np >= 2 ? 1 : 0;
}

int wgrad_split_launch(const WgradLaunch& a, hipStream_t stream) {
  int CI, CO;
  wgrad_split_tiles(a.cin_p, a.cout_p, &CI, &CO);
  if (a.cin_pad % CI || a.cout_pad % CO || a.cin_p % 4 || a.cout_p % 4 || a.ldx % 4 || a.lddz % 4 || a.H < 2 || a.W < 2) {
    set_error("wgrad_split: bad geometry");
    return MIMO_ERR_INVALID;
  }
  const bool ws = wgrad_use_ws(CI, CO);
  if (a.np == 2 && !(ws && a.store == 0 && a.dz_absmax && a.dz_absmax_n > 0 && a.dz_absmax_n <= kDzMaxSlots)) {
    set_error("wgrad_split: two MFMAs per product need the wave-specialised kernel, fp32 storage and WgradLaunch::dz_absmax");
    return MIMO_ERR_INVALID;
  }
  if (a.in_scale && !(ws && a.store == 0 && a.np >= 2 && a.in_shift)) {
    set_error("wgrad_split: this geometry cannot apply the input BatchNorm + ReLU in its loader (wgrad_split_fuses_input)");
    return MIMO_ERR_INVALID;
  }
  const int tilesY = ceil_div(a.H, ws ? wgrad_ws_tr(CI, a.store) : kWgTR), tilesX = ceil_div(a.W, kWgTC);
  const int numTiles = a.N * tilesY * tilesX;
  dim3 grid((a.cin_pad / CI) * (a.cout_pad / CO), a.splits);
  const int om = a.store;  // operand storage (WgradLaunch::store), see wg_mfma
  if (om < 0 || om > 4 || (om != 0 && a.np != 1) || ((om == 1 || om == 2) && (a.ldx % 8 || a.cin_p % 8)) ||
      (om != 0 && (a.lddz % 8 || a.cout_p % 8))) {
    set_error("wgrad_split: bad operand storage mode %d", om);
    return MIMO_ERR_INVALID;
  }
  if (ws) {
    grid = dim3(grid.x * grid.y);  // 1-D, decoded XCD-aware inside the kernel
#define WS_LAUNCH3(NP_, NI_, CI_, TR_, OM_)                                                                             \
  if constexpr ((NP_) >= 2 && (OM_) == 0) {                                                                             \
    if (a.in_scale)                                                                                                     \
      hipLaunchKernelGGL((wgrad_split_ws_kernel<NP_, NI_, CI_, TR_, OM_, true>), grid, dim3(512), 0, stream, a, tilesY, tilesX, numTiles); \
    else                                                                                                                \
      hipLaunchKernelGGL((wgrad_split_ws_kernel<NP_, NI_, CI_, TR_, OM_, false>), grid, dim3(512), 0, stream, a, tilesY, tilesX, numTiles); \
  } else {                                                                                                              \
    hipLaunchKernelGGL((wgrad_split_ws_kernel<NP_, NI_, CI_, TR_, OM_, false>), grid, dim3(512), 0, stream, a, tilesY, tilesX, numTiles); \
  }
#define WS_LAUNCH2(NI_, CI_, TR_)                \
  switch (om) {                                  \
    case 1: WS_LAUNCH3(1, NI_, CI_, 4, 1); break; /* both operands 16-bit: 4-row tiles */ \
    case 2: WS_LAUNCH3(1, NI_, CI_, 4, 2); break; \
    case 3: WS_LAUNCH3(1, NI_, CI_, TR_, 3); break; \
    case 4: WS_LAUNCH3(1, NI_, CI_, TR_, 4); break; \
    default:                                     \
      if (a.np == 1) {                           \
        WS_LAUNCH3(1, NI_, CI_, TR_, 0);         \
      } else if (a.np == 2) {                    \
        WS_LAUNCH3(2, NI_, CI_, TR_, 0);         \
      } else {                                   \
        WS_LAUNCH3(3, NI_, CI_, TR_, 0);         \
      }                                          \
  }
#define WS_LAUNCH(NI_)     \
  if (CI == 64) {          \
    WS_LAUNCH2(NI_, 64, 2); \
  } else {                 \
    WS_LAUNCH2(NI_, 32, 4); \
  }
    switch (CO) {
      case 32: WS_LAUNCH(2); break;
      case 48: WS_LAUNCH(3); break;
      default: WS_LAUNCH(4); break;
    }
#undef WS_LAUNCH3
#undef WS_LAUNCH2
#undef WS_LAUNCH
    MIMO_KERNEL_CHECK();
    return MIMO_OK;
  }
#define WG_LAUNCH3(MI, NI, WM, WN, WK, NP_, OM_) \
  hipLaunchKernelGGL((wgrad_split_kernel<MI, NI, WM, WN, WK, NP_, OM_>), grid, dim3(256), 0, stream, a, tilesY, tilesX, numTiles)
#define WG_LAUNCH(MI, NI, WM, WN, WK)               \
  switch (om) {                                     \
    case 1: WG_LAUNCH3(MI, NI, WM, WN, WK, 1, 1); break; \
    case 2: WG_LAUNCH3(MI, NI, WM, WN, WK, 1, 2); break; \
    case 3: WG_LAUNCH3(MI, NI, WM, WN, WK, 1, 3); break; \
    case 4: WG_LAUNCH3(MI, NI, WM, WN, WK, 1, 4); break; \
    default:                                        \
      if (a.np == 1) {                              \
        WG_LAUNCH3(MI, NI, WM, WN, WK, 1, 0);       \
      } else {                                      \
        WG_LAUNCH3(MI, NI, WM, WN, WK, 3, 0);       \
      }                                             \
  }
  const int key = CI * 100 + CO;
  switch (key) {
    case 3232: WG_LAUNCH(2, 2, 1, 1, 4); break;
    case 3248: WG_LAUNCH(2, 3, 1, 1, 4); break;
    case 3264: WG_LAUNCH(2, 2, 1, 2, 2); break;
    case 4832: WG_LAUNCH(3, 2, 1, 1, 4); break;
    case 4848: WG_LAUNCH(3, 3, 1, 1, 4); break;
    case 4864: WG_LAUNCH(3, 2, 1, 2, 2); break;
    case 6432: WG_LAUNCH(2, 2, 2, 1, 2); break;
    case 6448: WG_LAUNCH(2, 3, 2, 1, 2); break;
    default: WG_LAUNCH(2, 2, 2, 2, 1); break;
  }
#undef WG_LAUNCH3
#undef WG_LAUNCH
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

}  // namespace mimo
