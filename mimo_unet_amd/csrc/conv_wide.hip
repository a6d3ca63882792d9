// 3x3 convolution (forward / data gradient), split16 arithmetic, "wide" decomposition (round 3).
//
// Same arithmetic as conv_bf16x3.hip (a.b ~= a_hi.b_hi + a_hi.b_lo + a_lo.b_hi on the 16-bit MFMA, fp16 pairs in the
// forward, bf16 pairs in the data gradient, fp32 accumulation) on a decomposition that moves fewer bytes and issues
// fewer instructions per MFMA.  What the round-2 ablation found (profiles/r02/conv_ablation.txt): the 256-pixel x
// 64-channel kernel is bound by neither the MFMA pipe (50-56 % busy) nor HBM but by everything around the MFMAs — the
// staging waves re-stream 74 KB of weights per 46 KB input tile and 32-channel chunk, the lone MFMA wave of a SIMD reads
// 0.33 ds_read_b128 per MFMA and shares the SIMD's issue port (a 16x16x32 MFMA holds it for 8 of its 16 cycles) with
// the staging wave next to it, and there is a workgroup barrier every 108-144 MFMAs.  Here:
//   * tile = 512 output pixels x 32 / 64 output channels; K is walked in 16-channel chunks ("stages"), so the halo
//     tile of a stage is 612 x [hi 16 | lo 16] = 39 KB and two of them fit beside the weights.  Per MFMA cycle the
//     workgroup stages 37-53 % fewer bytes than the 256-pixel kernel (the weight image of a chunk serves twice the pixels).
//   * v_mfma_f32_32x32x16_{f16,bf16}: a consumer wave owns 128 pixels x NB channels = 4 x NF accumulator tiles of 32 x 32
//     (128 registers at NB = 64).  Half the MFMA instructions for the same flops (the issue port is held 8 of 32
//     cycles), 0.25 ds_read_b128 per 16-cycle MFMA equivalent, and K = 16 per instruction: channel counts pad to 16,
//     not 32 (45 -> 48, no tap pairing needed).
//   * weights by LDS-DMA into unpadded 64-byte rows (XOR swizzle on the per-lane SOURCE address): NB = 64: one tap row
//     (3 taps) per phase, ring of three buffers, DMA two phases ahead of its use; NB = 32: all nine taps of a chunk
//     per phase, two buffers.  One workgroup barrier per 72 (NB = 64) / 108 (NB = 32) MFMAs of 32 cycles, i.e. per
//     2304 / 3456 matrix-pipe cycles (256-pixel kernel: 1152-2304).
//   * input rows are 80 bytes (64 + 16 pad): ds_read_b128 by 16 consecutive pixels hits 16 distinct 16-byte slots
//     (5 is coprime to 16), offsets of the nine taps are immediates / phase scalars.
//   * accumulators transposed (weights = MFMA A operand): lane (pixel, half h) holds channels 8j + 4h .. +3, j = 0..3,
//     of each 32-channel tile -> 16-byte stores.  BatchNorm partial sums: a lane sees 16 channels per tile; adjacent
//     pixel lanes split them (one DPP reduce-scatter step per tile) so that the sums persist in 4 registers per tile
//     and moment instead of 16.
// Wave roles as in conv3x3_ws_kernel: waves 0-3 consume (ds_read + MFMA + epilogue), waves 4-7 stage (global loads two
// stages ahead in registers, fp32 -> fp16 hi/lo split or 16-byte copies of the pre-split dz, LDS-DMA of the weights);
// persistent workgroups, one per CU, XCD-aware order (channel tiles of a pixel tile share an L2).
// The 16-bit storage modes (MODE 4-7: bf16-mixed / 16-mixed forward and data gradient) run on the same kernel with one
// MFMA per product: a K chunk is 32 channels of plain 16-bit data in the same 64-byte rows, the loaders are 16-byte copies.
// Which layers run here: sched::wide_config (tile_sched.h) — the packer lays the weights out for the decomposition the
// launch will use (ConvLaunch::wide).  Measured (profiles/r03/): matrix pipe 62-69 % busy on the
// wide layers (256-pixel kernel 46-55 %), per-layer forward -7...-13 % in split16, -30...-40 % in the 16-bit modes.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "tile_sched.h"

#ifndef MIMO_WIDE_DEFER
#define MIMO_WIDE_DEFER 1  // 0: the per-tile epilogue runs between the tiles as in rounds 3-4 (A/B builds)
#endif
#define WD_DEFER_ENABLED (MIMO_WIDE_DEFER != 0)

namespace mimo {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;

__device__ __forceinline__ f32x16 mfma32(bf16x8_t a, bf16x8_t b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(f16x8_t a, f16x8_t b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// value of the lane with the lowest bit flipped (quad_perm [1, 0, 3, 2])
__device__ __forceinline__ float dpp_xor1(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
}

constexpr float kWideF16Scale = 256.f;  // forward weights are packed x 2^8 (as conv_bf16x3.hip)

constexpr int kXPitch = 80;             // bytes per input row: [hi 16 | lo 16] 16-bit values + 16 pad
constexpr int kMaxPix = sched::kWideMaxPix;
constexpr int kXBytes = kMaxPix * kXPitch;  // 51200

}  // namespace

// MODE 0: data gradient (bf16 pairs, input = pre-split dz records [hi rc | lo rc] per 32-channel chunk)
// MODE 1: forward (fp16 pairs, fp32 input split on the way into LDS; bias, BatchNorm sums, inference epilogue)
// MODE 4 / 5: bf16-mixed forward / data gradient; MODE 6 / 7: fp16-mixed (mimo_precision *_MIXED: activations and
//   gradients stored as plain NHWC 16-bit tensors).  One MFMA per product; the 64-byte LDS row then holds 32 channels
//   of ONE value each where the split modes hold [hi 16 | lo 16], a chunk is 32 channels = two K = 16 MFMA steps on
//   the row's two halves, the loaders are 16-byte copies, the output is rounded once in the epilogue.
// NF: 32-channel tiles per workgroup; TPP: taps per phase (3 = one tap row, 9 = a whole chunk)
// EPI (forward): the inference epilogue — eval-mode BatchNorm + ReLU (+ Dropout2d multipliers) folded into the store, no
// statistics; else bias + BatchNorm partial sums
// FIN (split16 forward, training / grad-enabled path): the input is the producing convolution's pre-activation tensor and the
// producers apply its BatchNorm + ReLU (ConvLaunch::in_scale / in_shift) in front of the fp16 split.  A thread's units all
// hold the same channel quad of a chunk (256 threads step over whole pixels), so a stage costs two 16-byte loads of constants
// per thread — issued one stage ahead, in front of that phase's weight DMA and input loads, so that the counted waits stay valid.
template <int NF, int MODE, int TPP, bool EPI, bool FIN = false>
__global__ __launch_bounds__(512, 2) void conv3x3_wide_kernel(ConvLaunch a, int TR, int TC, int tilesY, int tilesX,
                                                                int numTiles, int gx, int coTiles) {
  static_assert(MODE == 0 || MODE == 1 || (MODE >= 4 && MODE <= 7), "split16 and 16-bit storage modes");
  static_assert(!FIN || (MODE == 1 && !EPI), "fused input BatchNorm + ReLU: the split16 forward with statistics");
  static_assert((NF == 2 && TPP == 3) || (NF == 1 && TPP == 9), "instances: 64 channels x tap rows, 32 channels x chunks");
  constexpr bool FWD = MODE == 1 || MODE == 4 || MODE == 6;
  constexpr bool S16 = MODE >= 4;                 // 16-bit storage: plain 16-bit input and output, one MFMA per product
  constexpr bool F16 = MODE == 1 || MODE >= 6;    // element type fp16 (else bf16); fp16 forward weights carry x 2^8
  constexpr bool CVT = MODE == 1;                 // the loader splits fp32 input into fp16 (hi, lo)
  constexpr int CKC = S16 ? 32 : 16;              // input channels per chunk
  constexpr bool STATS = FWD && !EPI;
  static_assert(FWD || !EPI, "the inference epilogue belongs to the forward");
  typedef typename std::conditional<F16, f16x8_t, bf16x8_t>::type V8;
  typedef typename std::conditional<F16, _Float16, __bf16>::type ET;
  constexpr int NB = NF * 32;
  constexpr int MF = 4;                      // 32-pixel fragments per consumer wave
  constexpr int SM = NF == 1 ? 2 : 1;        // fragments per software-pipeline slot (6 MFMAs per slot)
  constexpr int NSLOT = MF / SM;
  constexpr int PARTS = 9 / TPP;             // phases per stage
  constexpr int XU = 10;                     // 16-byte units of an input tile per producer thread (640 x 4 / 256)
  // units [XK0(R), XK0(R + 1)) are staged in phase R of a stage: 4 + 4 + 2 over three phases, all ten in one
#define XK0(R) (PARTS == 1 ? ((R) <= 0 ? 0 : XU) : ((R) <= 0 ? 0 : (R) == 1 ? 4 : (R) == 2 ? 8 : XU))
#define XKN(R) (XK0((R) + 1) - XK0(R)) /* loads of phase R */
  constexpr int WUNITS = TPP * NB * 4;       // 16-byte units of a phase's weights
  constexpr int WU = (WUNITS + 255) / 256;   // DMA instructions per producer wave and phase (the same for every wave)
  constexpr int WPHB = WU * 256 * 16;        // bytes per weight buffer (rounded up to whole DMA rounds)
  constexpr int NWB = TPP == 3 ? 3 : 2;      // weight buffers
  constexpr int EPIB = FWD ? 3 * NB * 4 : 0;  // per-channel epilogue constants: bias, inference scale, shift
  // deferred epilogue (see the consumers): the lane-private BatchNorm partial sums live in LDS (16 floats per consumer
  // lane), not in 16 registers next to the 64 saved accumulators
  constexpr bool DEFER = NF == 1 && MODE == 0 && !EPI && WD_DEFER_ENABLED;  // (the forward: measured slower, see DEFER below)
  constexpr int SUMB = (DEFER && FWD) ? 4 * 2 * 256 * 8 : 0;
  static_assert(2 * kXBytes + NWB * WPHB + EPIB + SUMB <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * kXBytes + NWB * WPHB + EPIB + SUMB];
  unsigned char* const xs = lds;
  unsigned char* const ws = lds + 2 * kXBytes;
  float* const epi = reinterpret_cast<float*>(lds + 2 * kXBytes + NWB * WPHB);

#ifdef MIMO_WIDE_ABLATE
  // timing-only builds (-DMIMO_WIDE_ABLATE=bits, results are wrong): 1 = producers skip the input-tile loads, 2 = the
  // input-tile LDS stores, 4 = the weight DMA; 8 = consumers skip the MFMAs, 16 = the fragment reads, 32 = the per-tile
  // epilogue (the accumulators stay live); 64 = every input tile addressed as an interior one, 128 = ... and read
  // from the first tile's addresses (cache-hot)
  constexpr int abl = MIMO_WIDE_ABLATE;
#else
  constexpr int abl = 0;
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int TCP = TC + 2, TRP = TR + 2;
  const int npix_lds = TRP * TCP, npix_out = TR * TC;
  const int nchunks = (a.cin_p + CKC - 1) / CKC;
  const int v_ = xcd_virtual_index((int)blockIdx.x, gx * coTiles);
  const int vbx = v_ / coTiles;
  const int co0 = (v_ - vbx * coTiles) * NB;
  const int ntiles_mine = vbx < numTiles ? (numTiles - 1 - vbx) / gx + 1 : 0;
  const int nstages = ntiles_mine * nchunks;
  const int nphases = PARTS * nstages;
  const int rows_pad = coTiles * NB;  // packed weight rows per tap

  if (FWD) {
    // The epilogue's per-channel constants come from LDS: a global load in the epilogue would be waited for with
    // vmcnt(0), which on this chip also waits for every output store in flight (stores count in vmcnt)
    if (tid < NB) {
      const int c = co0 + tid;
      epi[tid] = c < a.cout_pad ? a.bias[c] : 0.f;
      if (EPI) {
        epi[NB + tid] = c < a.cout_store ? a.ep_scale[c] : 0.f;
        epi[2 * NB + tid] = c < a.cout_store ? a.ep_shift[c] : 0.f;
      }
    }
    __syncthreads();
  }

  if (wave >= 4) {
    // =============================== producers ===============================
    // The staging waves share their SIMD's issue port with an MFMA wave, so what they cost is their INSTRUCTION count
    // (timing builds: with the MFMAs compiled out a phase of the first version took 0.95 us of producer instructions —
    // per-phase scalar division chains decoding the stage, ~20 vector instructions of reflection / bounds arithmetic
    // per 16-byte unit).  Everything that depends only on the tile is therefore computed once per tile: the byte offset
    // of each unit's source pixel from the image base (reflected / clipped), its validity bit, the image base; a stage
    // adds the chunk's byte offset.  The stage / phase bookkeeping is incremental (no divisions in the loop).
    const int ptid = tid - 256;
    f32x4 xreg[XU];
    const unsigned ws_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ws;
    // weight unit u = ptid + 256 k of a phase: LDS row u >> 2 = tap * NB + channel, slot u & 3 <- source slot
    // (u & 3) ^ ((channel >> 2) & 3).  Units past the phase's image (WUNITS not a multiple of 256) re-read its
    // last unit into the buffer's slack.
    int wsrc[WU];
#pragma unroll
    for (int k = 0; k < WU; ++k) {
      const int u = min(ptid + k * 256, WUNITS - 1), row = u >> 2;
      const int tp = row / NB, ch = row - tp * NB;
      wsrc[k] = ((tp * rows_pad + ch) * 4 + ((u & 3) ^ ((ch >> 2) & 3))) * 16;  // bytes from the phase's first row
    }
    const unsigned wdst0 = __builtin_amdgcn_readfirstlane(ws_lds + (unsigned)((ptid & ~63) * 16));  // this wave's 1 KB
    int u_rc[XU];   // halo-tile coordinates of the thread's units: (row << 16) | column
    int u_off[XU];  // current load tile: byte offset of the unit's source from the image base (chunk 0)
    unsigned u_val = 0;  // ... bit k: the source pixel lies inside the image (data gradient; the forward reflects)
#pragma unroll
    for (int k = 0; k < XU; ++k) {
      const int p = min((ptid + k * 256) >> 2, npix_lds - 1);
      const int tr = p / TCP, tc = p - tr * TCP;
      u_rc[k] = (tr << 16) | tc;
      u_off[k] = 0;
    }
    const int uq = ptid & 3;  // 16-byte slot of the thread's units within the LDS row (u & 3: 256 is a multiple of 4)
    // byte offset of the unit within its pixel, full chunks: forward 4 channels x 4 B per slot; data gradient: 32-channel
    // record [hi 32 | lo 32] bf16, the chunk's half adds 32 B per stage
    const int uq_off = (CVT || S16) ? 16 * uq : (uq >> 1) * 64 + (uq & 1) * 16;
    const int esz = S16 ? 2 : 4;  // bytes per element of the input tensor (ldx counts elements)
    const char* const wpk = reinterpret_cast<const char*>(a.wpk) + (size_t)co0 * 64;
    const int wph_bytes = TPP * rows_pad * 64;  // weight bytes between consecutive phases of a tile
    // ---- load stream: the stage whose input tile is being fetched (two stages ahead of the consumers) ----
    int l_stage = 0, l_ck = 0, l_ti = 0;
    const char* l_img = nullptr;  // image base of the load tile
    int l_y0 = 0, l_x0 = 0;
    bool l_new = true;            // the stage is the first of its tile: unit offsets are recomputed part by part
    // FIN: BatchNorm scale / shift of the thread's channel quad — of the stage being stored (in_sc / in_sh) and of the stage
    // being loaded (in_sc_n / in_sh_n); channels past cin_p come from the zero page: relu(0 * 0 + 0) keeps them zero
    f32x4 in_sc = f32x4{0.f, 0.f, 0.f, 0.f}, in_sh = in_sc, in_sc_n = in_sc, in_sh_n = in_sc;
#define WD_LOAD_SS(SC, SH, CK)                                                                       \
  if (FIN) {                                                                                         \
    const int c_ = (CK) * CKC + 4 * uq;                                                              \
    /* UNCONDITIONAL: in_scale / in_shift are allocated with kFinSlack zero floats behind the layer's channels (plan.hip), \
       so a chunk's quads past cin_p read zeros (relu(0 * 0 + 0) keeps those channels zero) and the stage always issues      \
       exactly the two loads the counted waits below assume — no select the compiler could turn into a branch */ \
    SC = *reinterpret_cast<const f32x4*>(a.in_scale + c_);                                           \
    SH = *reinterpret_cast<const f32x4*>(a.in_shift + c_);                                           \
  }
#define WD_TILE(TI)                                                                                  \
  {                                                                                                  \
    int t_ = vbx + (TI) * gx;                                                                        \
    const int tx_ = t_ % tilesX;                                                                     \
    t_ /= tilesX;                                                                                    \
    const int ty_ = t_ % tilesY;                                                                     \
    const int n_ = t_ / tilesY;                                                                      \
    l_y0 = ty_ * TR - a.off;                                                                         \
    l_x0 = tx_ * TC - a.off;                                                                         \
    l_img = reinterpret_cast<const char*>(a.x) + (size_t)n_ * a.Hi * a.Wi * a.ldx * esz;             \
  }
    // unit offsets K0..K1 of the load tile (first stage of a tile)
#define WD_OFFS(K0, K1)                                                                              \
  _Pragma("unroll") for (int k_ = (K0); k_ < (K1); ++k_) {                                           \
    int iy_ = l_y0 + (u_rc[k_] >> 16), ix_ = l_x0 + (u_rc[k_] & 0xffff);                             \
    if (FWD) { /* reflect padding */                                                                 \
      iy_ = max(iy_, -iy_);                                                                          \
      iy_ = max(min(iy_, 2 * a.Hi - 2 - iy_), 0);                                                    \
      ix_ = max(ix_, -ix_);                                                                          \
      ix_ = max(min(ix_, 2 * a.Wi - 2 - ix_), 0);                                                    \
    } else { /* zero outside the image (transposed convolution) */                                   \
      const bool in_ = iy_ >= 0 && iy_ < a.Hi && ix_ >= 0 && ix_ < a.Wi;                             \
      u_val = in_ ? u_val | (1u << k_) : u_val & ~(1u << k_);                                        \
      iy_ = in_ ? iy_ : 0;                                                                           \
      ix_ = in_ ? ix_ : 0;                                                                           \
    }                                                                                                \
    u_off[k_] = (iy_ * a.Wi + ix_) * a.ldx * esz + uq_off;                                           \
  }
    // input units K0..K1 of the load stage -> registers.  Masked-out units are LOADED from the zero page (common.h),
    // never selected after the load.  A short last chunk takes the path with per-unit channel tests.
#define WD_LOAD_X(K0, K1)                                                                            \
  if (abl & 1) {                                                                                     \
    _Pragma("unroll") for (int k_ = (K0); k_ < (K1); ++k_) xreg[k_] = f32x4{1.f, 2.f, 3.f, 4.f};     \
  } else {                                                                                           \
    if (l_new) {                                                                                     \
      WD_OFFS(K0, K1)                                                                                \
    }                                                                                                \
    const int rc_ = min(32, a.cin_p - (l_ck >> 1) * 32); /* split16 data gradient: channels of the 32-channel record */ \
    /* a unit = 16 bytes of one pixel: 4 fp32 channels (split16 forward), 8 16-bit channels (16-bit storage), 8 bf16  \
       values of a pair record's hi or lo half (split16 data gradient) */                              \
    const bool full_ = (CVT || S16) ? l_ck * CKC + CKC <= a.cin_p : rc_ == 32;                       \
    const char* b_ = l_img + ((CVT || S16) ? l_ck * 64 : (l_ck >> 1) * 128 + (l_ck & 1) * 32);       \
    if (full_) {                                                                                     \
      _Pragma("unroll") for (int k_ = (K0); k_ < (K1); ++k_) {                                       \
        if (FWD)                                                                                     \
          xreg[k_] = *reinterpret_cast<const f32x4*>(b_ + u_off[k_]);                                \
        else                                                                                         \
          xreg[k_] = *reinterpret_cast<const f32x4*>(                                                \
              (u_val >> k_) & 1 ? b_ + u_off[k_] : reinterpret_cast<const char*>(kZeroPage));        \
      }                                                                                              \
    } else {                                                                                         \
      _Pragma("unroll") for (int k_ = (K0); k_ < (K1); ++k_) {                                       \
        if (CVT || S16) {                                                                            \
          const bool ok_ = (FWD || ((u_val >> k_) & 1)) && l_ck * CKC + (CKC / 4) * uq < a.cin_p;    \
          xreg[k_] = *reinterpret_cast<const f32x4*>(ok_ ? b_ + u_off[k_] : reinterpret_cast<const char*>(kZeroPage)); \
        } else { /* record [hi rc | lo rc]: the lo half sits rc, not 32, elements behind the hi half */ \
          const int e_ = (l_ck & 1) * 16 + (uq & 1) * 8;                                             \
          const bool ok_ = ((u_val >> k_) & 1) && e_ < rc_;                                          \
          xreg[k_] = *reinterpret_cast<const f32x4*>(                                                \
              ok_ ? b_ + u_off[k_] + (uq >> 1) * (rc_ - 32) * 2 : reinterpret_cast<const char*>(kZeroPage)); \
        }                                                                                            \
      }                                                                                              \
    }                                                                                                \
  }
    // the load stream moves on to the next stage (stays on the last one past the end)
#define WD_NEXT_STAGE()                                                                              \
  {                                                                                                  \
    l_new = false;                                                                                   \
    if (l_stage + 1 < nstages) {                                                                     \
      ++l_stage;                                                                                     \
      if (++l_ck == nchunks) {                                                                       \
        l_ck = 0;                                                                                    \
        ++l_ti;                                                                                      \
        l_new = true;                                                                                \
        WD_TILE(l_ti)                                                                                \
      }                                                                                              \
    }                                                                                                \
  }
#define WD_STORE_X(K0, K1, BUF)                                                                      \
  if (abl & 2) {                                                                                     \
    _Pragma("unroll") for (int k_ = (K0); k_ < (K1); ++k_) asm volatile("" ::"v"(xreg[k_]));         \
  } else _Pragma("unroll") for (int k_ = (K0); k_ < (K1); ++k_) {                                    \
    const int p_ = (ptid + k_ * 256) >> 2;                                                           \
    /* UNCONDITIONAL: rows [npix_lds, 640) are slack nobody reads.  A store skipped by a branch leaves its load    \
       un-waited on that path, and hipcc then drains the whole queue (vmcnt(0)) before it reuses the register —   \
       every phase, which collapsed the prefetch depth to zero (found in the ISA) */                              \
    {                                                                                                \
      f32x4 v_ = xreg[k_];                                                                           \
      if (FIN) {                                                                                     \
        v_[0] = fmaxf(fmaf(v_[0], in_sc[0], in_sh[0]), 0.f);                                         \
        v_[1] = fmaxf(fmaf(v_[1], in_sc[1], in_sh[1]), 0.f);                                         \
        v_[2] = fmaxf(fmaf(v_[2], in_sc[2], in_sh[2]), 0.f);                                         \
        v_[3] = fmaxf(fmaf(v_[3], in_sc[3], in_sh[3]), 0.f);                                         \
      }                                                                                              \
      unsigned char* row_ = xs + (BUF) * kXBytes + p_ * kXPitch;                                     \
      if (CVT) {                                                                                     \
        f16x4_t hi_, lo_;                                                                            \
        hi_[0] = (_Float16)v_[0];                                                                    \
        hi_[1] = (_Float16)v_[1];                                                                    \
        hi_[2] = (_Float16)v_[2];                                                                    \
        hi_[3] = (_Float16)v_[3];                                                                    \
        lo_[0] = (_Float16)(v_[0] - (float)hi_[0]);                                                  \
        lo_[1] = (_Float16)(v_[1] - (float)hi_[1]);                                                  \
        lo_[2] = (_Float16)(v_[2] - (float)hi_[2]);                                                  \
        lo_[3] = (_Float16)(v_[3] - (float)hi_[3]);                                                  \
        *reinterpret_cast<f16x4_t*>(row_ + uq * 8) = hi_;                                            \
        *reinterpret_cast<f16x4_t*>(row_ + 32 + uq * 8) = lo_;                                       \
      } else {                                                                                       \
        *reinterpret_cast<f32x4*>(row_ + uq * 16) = v_;                                              \
      }                                                                                              \
    }                                                                                                \
  }
    // ---- weight stream: phases in consumer order; w_src walks the packed image of the channel tile, back to its start
    // at every tile.  LDS-DMA by inline asm, not the builtin: with a DMA it knows of in flight hipcc waits vmcnt(0) in
    // front of every use of an ordinary load (conv_bf16x3.hip WS_DMA_W).  M0 = LDS byte address of the wave's first
    // lane, saved and restored inside the statement.
    int w_ph = 0, w_pht = 0;  // phase index of the next DMA, its index within the tile
    const char* w_src = wpk;
    const int pht_n = PARTS * nchunks;
#define WD_DMA_W(BUF)                                                                                \
  {                                                                                                  \
    if (!(abl & 4)) {                                                                                \
      _Pragma("unroll") for (int k_ = 0; k_ < WU; ++k_) {                                            \
        const unsigned dst_ = wdst0 + (unsigned)((BUF) * WPHB + k_ * 4096);                          \
        unsigned keep_;                                                                              \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(w_src + wsrc[k_]), "s"(dst_) : "memory");                  \
      }                                                                                              \
    }                                                                                                \
    if (w_ph + 1 < nphases) { /* past the end: the last phase's weights again, into a buffer nobody reads */ \
      ++w_ph;                                                                                        \
      w_src += wph_bytes;                                                                            \
      if (++w_pht == pht_n) {                                                                        \
        w_pht = 0;                                                                                   \
        w_src = wpk;                                                                                 \
      }                                                                                              \
    }                                                                                                \
  }
    // all vector-memory operations but the N youngest are done; LDS stores retired; phase barrier
#define WD_WAIT_BAR(N) asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"((abl & 5) ? 0 : (N)) : "memory");

    // ---- prologue: stage 0 in input buffer 0, weights of phase 0 (and 1) on their way; input of stage 1 in registers
    WD_TILE(0)
    WD_LOAD_SS(in_sc, in_sh, 0)
    WD_LOAD_X(0, XU)
    WD_DMA_W(0)
    if (NWB == 3) {
      WD_DMA_W(1)
    }
    WD_STORE_X(0, XU, 0)
    WD_NEXT_STAGE()
    WD_LOAD_SS(in_sc_n, in_sh_n, l_ck)
    WD_LOAD_X(0, XU)
    // Every load of the prologue is waited for HERE, visibly to the compiler: the loop's first consumer of a register
    // is otherwise waited for with the count that is safe on the path from the prologue too — vmcnt(0) if that
    // register was the prologue's last load — once per loop iteration (found in the ISA)
    _Pragma("unroll") for (int k_ = 0; k_ < XU; ++k_) asm volatile("" ::"v"(xreg[k_]));
    WD_WAIT_BAR(0)
    // ---- phase (j, R): store part R of stage j + 1 into the input buffer the consumers released at the start of stage
    // j, start the DMA of a later phase's weights into the buffer released at the last barrier, refill the registers
    // with part R of stage j + 2.  Straight-line: past the end the loads re-read the last stage and the stores / DMAs
    // go to buffers nobody reads any more (exact vmcnt counts need unconditional code).
    int wslot = NWB == 3 ? 2 : 1;  // buffer of the next DMA = (ph + NWB - 1) % NWB
    for (int j = 0; j < nstages; ++j) {
      const int xb = (j + 1) & 1;
      if (FIN) { /* the constants of stage j + 1 (loaded one stage ago, older than every load still in flight) */
        in_sc = in_sc_n;
        in_sh = in_sh_n;
      }
      WD_NEXT_STAGE()  // the load stream: stage j + 2
#define WD_PHASE(R)                                                                                  \
  {                                                                                                  \
    if ((R) == 0) {                                                                                  \
      WD_LOAD_SS(in_sc_n, in_sh_n, l_ck) /* in front of this phase's DMA and input loads */          \
    }                                                                                                \
    WD_STORE_X(XK0(R), XK0((R) + 1), xb)                                                             \
    WD_DMA_W(wslot)                                                                                  \
    wslot = wslot + 1 == NWB ? 0 : wslot + 1;                                                        \
    WD_LOAD_X(XK0(R), XK0((R) + 1))                                                                  \
    /* three buffers: the weights of phase ph + 1 were issued one phase ago, in front of that phase's input loads     \
       (FIN, phase 0: the two loads of constants sit between them and this phase's DMA) */                            \
    WD_WAIT_BAR(NWB == 3 ? XKN(((R) + 2) % 3) + WU + XKN(R) + ((FIN && (R) == 0) ? 2 : 0) : XKN(R))  \
  }
      WD_PHASE(0)
      if (PARTS == 3) {
        WD_PHASE(1)
        WD_PHASE(2)
      }
#undef WD_PHASE
    }
#undef WD_LOAD_SS
#undef WD_TILE
#undef WD_OFFS
#undef WD_NEXT_STAGE
#undef WD_LOAD_X
#undef WD_STORE_X
#undef WD_DMA_W
#undef WD_WAIT_BAR
#undef XK0
#undef XKN
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // DMAs past the end still target this workgroup's LDS
    if (STATS) __syncthreads();                       // the consumers combine their BatchNorm sums through LDS
    return;
  }

  // =============================== consumers ===============================
  const int px = lane & 31, h = lane >> 5;
  int pbase[MF], prc[MF];
#pragma unroll
  for (int m = 0; m < MF; ++m) {
    const int idx = (wave * MF + m) * 32 + px;
    const int r = idx / TC, c = idx - r * TC;
    pbase[m] = idx < npix_out ? (r * TCP + c) * kXPitch + h * 16 : h * 16;
    prc[m] = idx < npix_out ? (r << 16) | c : (0x4000 << 16);  // tile-relative (row, column); 0x4000 = not in the tile
  }
  // weight fragment (MFMA A operand): lane (row px of the 32-channel tile, K half h): hi at slot h, lo at slot 2 + h,
  // slots XOR-swizzled by (row >> 2) & 3
  const int wsw = (px >> 2) & 3;
  const int w_hi = px * 64 + ((h ^ wsw) << 4), w_lo = px * 64 + (((2 + h) ^ wsw) << 4);
  const int row1 = TCP * kXPitch;  // input-tile byte offset of one tap row

  f32x16 acc[MF][NF];
#pragma unroll
  for (int m = 0; m < MF; ++m)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][nf][i] = 0.f;
  // BatchNorm partial sums over all tiles of this persistent workgroup.  Lane (px, h) holds, of channel quad j of tile
  // nf (channels 8 j + 4 h .. + 3), the two channels 2 (px & 1) .. + 1 — summed over its own and its neighbour's pixels
  f32x2 s1[NF][4], s2[NF][4];  // (DEFER: in LDS until the end, `lsum`)
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s1[nf][j] = f32x2{0.f, 0.f};
      s2[nf][j] = f32x2{0.f, 0.f};
    }
  typedef typename std::conditional<S16, ET, float>::type OT_;  // element type of the output tensor
  const float winv = !F16 ? 1.f : a.wmax ? w16_scale(*a.wmax, true) : 1.f / kWideF16Scale;  // the fp16 weight image's scale
  (void)winv;
  const bool odd = px & 1;
  // DEFER (round 5; 32-channel instances of the split16 training forward / data gradient): the epilogue of tile t does not
  // run between the tiles — where the four consumer waves of every CU push 64 KB each at the same moment and the matrix
  // pipe idles until the burst has drained (profiles/r04/thin_role_ablation.txt: 19-25 % of a thin layer) — but as 16
  // pieces (one 16-byte store each) inside the first phase of tile t + 1, one per slot, interleaved with that slot's
  // MFMAs: the finished accumulators move to `sav`, the new tile's first MFMAs start from a zero C operand.  Same
  // values, same order of the BatchNorm sums: bit-identical results.
  // The partial BatchNorm sums of a lane live in LDS (`lsum`, 16 floats per lane) instead of 16 registers.
  f32x2* const lsum = reinterpret_cast<f32x2*>(lds + 2 * kXBytes + NWB * WPHB + EPIB) + (tid & 255);  // [quad][moment][lane]
  if (SUMB) {
#pragma unroll
    for (int i = 0; i < 8; ++i) lsum[i * 256] = f32x2{0.f, 0.f};
  }
  // The stores are BUFFER stores on a descriptor of the tile's image: a lane whose pixel lies outside the image (or a
  // padded channel quad) carries an offset past the descriptor's range and the hardware drops its store — no branch, no
  // address select, and the offsets are 32-bit.
  f32x16 sav[DEFER ? MF : 1];
  unsigned sv_off[DEFER ? MF : 1];  // byte offset of the lane's pixel of fragment m (channel quad 0) in the image, or kNoStore
  __amdgpu_buffer_rsrc_t sv_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, 0, 0x00020000);
  constexpr unsigned kNoStore = 0x7fff0000u;  // beyond any image the launch accepts (conv3x3_wide_launch)
  f32x4 ep_b = f32x4{0.f, 0.f, 0.f, 0.f}, ep_t4 = ep_b, ep_q4 = ep_b;
  (void)sav; (void)sv_off; (void)sv_rsrc; (void)ep_b; (void)ep_t4; (void)ep_q4;

  const f32x16 kZero16 = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  (void)kZero16;
  V8 af[2][SM][2], wf[2][NF][2];  // [register set][fragment][hi, lo]

  // weights of tap T (within the phase) -> register set WS_
#define WC_READ_W(WS_, T)                                                                            \
  if (!(abl & 16)) _Pragma("unroll") for (int nf = 0; nf < NF; ++nf) {                                                \
    wf[WS_][nf][0] = *reinterpret_cast<const V8*>(wb_ + ((T) * NB + nf * 32) * 64 + w_hi);           \
    wf[WS_][nf][1] = *reinterpret_cast<const V8*>(wb_ + ((T) * NB + nf * 32) * 64 + w_lo);           \
  }
  // pixel fragments of slot S, tap T -> register set AS_.  Tap T of a phase: TPP = 3: (tap row of the phase, T);
  // TPP = 9: (T / 3, T % 3)
#define WC_READ_A(AS_, T, S)                                                                         \
  if (!(abl & 16)) _Pragma("unroll") for (int i = 0; i < SM; ++i) {                                                   \
    const unsigned char* p_ = xb_ + pbase[(S) * SM + i] +                                            \
                              (TPP == 3 ? ro_ + (T) * kXPitch : ((T) / 3) * row1 + ((T) % 3) * kXPitch); \
    af[AS_][i][0] = *reinterpret_cast<const V8*>(p_);                                                \
    af[AS_][i][1] = *reinterpret_cast<const V8*>(p_ + 32);                                           \
  }
#define WC_MFMA(AS_, WS_, S) WC_MFMA_Z(AS_, WS_, S, false)
  // Z: the first MFMA of each accumulator tile starts from a zero C operand (first tap of a tile in the DEFER path)
#define WC_MFMA_Z(AS_, WS_, S, Z)                                                                    \
  if (abl & 8) {                                                                                     \
    _Pragma("unroll") for (int i = 0; i < SM; ++i) {                                                 \
      asm volatile("" ::"v"(af[AS_][i][0]), "v"(af[AS_][i][1]));                                     \
    }                                                                                                \
    _Pragma("unroll") for (int nf = 0; nf < NF; ++nf) {                                              \
      asm volatile("" ::"v"(wf[WS_][nf][0]), "v"(wf[WS_][nf][1]));                                   \
    }                                                                                                \
  } else _Pragma("unroll") for (int i = 0; i < SM; ++i)                                              \
    _Pragma("unroll") for (int nf = 0; nf < NF; ++nf) {                                              \
      if (S16) { /* the row's halves are channels 0-15 and 16-31 of the chunk */                     \
        acc[(S) * SM + i][nf] = mfma32(wf[WS_][nf][1], af[AS_][i][1], acc[(S) * SM + i][nf]);        \
      } else { /* (hi, lo) pairs: lo.hi + hi.lo + hi.hi */                                           \
        acc[(S) * SM + i][nf] = mfma32(wf[WS_][nf][0], af[AS_][i][1], (Z) ? kZero16 : acc[(S) * SM + i][nf]); \
        acc[(S) * SM + i][nf] = mfma32(wf[WS_][nf][1], af[AS_][i][0], acc[(S) * SM + i][nf]);        \
      }                                                                                              \
      acc[(S) * SM + i][nf] = mfma32(wf[WS_][nf][0], af[AS_][i][0], acc[(S) * SM + i][nf]);          \
    }
  // order of a slot pinned for the scheduler: its LDS reads (of the NEXT slot's fragments), then its MFMAs — left
  // alone hipcc sinks every read to just before its first use (one register set) and the lone MFMA wave of the SIMD
  // eats the LDS latency per fragment
#define WC_PIN(NREADS)                                                                               \
  __builtin_amdgcn_sched_group_barrier(0x100, (NREADS), 0);                                          \
  __builtin_amdgcn_sched_group_barrier(0x008, (S16 ? 2 : 3) * SM * NF, 0);
  // bias, store, BatchNorm partial sums of tile TI (accumulators complete), then clear them
  typedef typename std::conditional<S16, ET, float>::type OT;  // element type of the output tensor
  typedef ET et4 __attribute__((ext_vector_type(4)));
#define WC_EPILOGUE(TI)                                                                              \
  {                                                                                                  \
    int t_ = vbx + (TI) * gx;                                                                        \
    const int tx_ = t_ % tilesX;                                                                     \
    t_ /= tilesX;                                                                                    \
    const int ty_ = t_ % tilesY;                                                                     \
    const int n_ = t_ / tilesY;                                                                      \
    const int y0_ = ty_ * TR, x0_ = tx_ * TC;                                                        \
    OT* yimg_ = reinterpret_cast<OT*>(a.y) + (size_t)n_ * a.Ho * a.Wo * a.ldy + co0 + 4 * h;         \
    int yo_[MF]; /* element offset of the lane's pixel in the image; < 0: not stored */              \
    _Pragma("unroll") for (int m = 0; m < MF; ++m) {                                                 \
      const int oy = y0_ + (prc[m] >> 16), ox = x0_ + (prc[m] & 0xffff);                             \
      yo_[m] = (oy < a.Ho && ox < a.Wo) ? (oy * a.Wo + ox) * a.ldy : -1;                             \
      if (abl & 32) yo_[m] = (acc[m][0][0] + acc[m][NF - 1][15] == 12345.678f) ? yo_[m] : -1;        \
    }                                                                                                \
    _Pragma("unroll") for (int nf = 0; nf < NF; ++nf)                                                \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                \
        const int c_ = co0 + nf * 32 + 8 * j + 4 * h;                                                \
        f32x4 b_ = f32x4{0.f, 0.f, 0.f, 0.f}, esc_ = b_, esh_ = b_, emk_ = f32x4{1.f, 1.f, 1.f, 1.f}; \
        if (FWD) b_ = *reinterpret_cast<const f32x4*>(epi + c_ - co0);                               \
        if (EPI) {                                                                                   \
          esc_ = *reinterpret_cast<const f32x4*>(epi + NB + c_ - co0);                               \
          esh_ = *reinterpret_cast<const f32x4*>(epi + 2 * NB + c_ - co0);                           \
          if (a.ep_mask && c_ < a.cout_store)                                                        \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                         \
              if (c_ + i_ < a.ep_mask_ld) emk_[i_] = a.ep_mask[(size_t)n_ * a.ep_mask_ld + c_ + i_]; \
        }                                                                                            \
        f32x4 t4_ = f32x4{0.f, 0.f, 0.f, 0.f}, q4_ = t4_;                                            \
        _Pragma("unroll") for (int m = 0; m < MF; ++m) {                                             \
          f32x4 v = f32x4{acc[m][nf][4 * j], acc[m][nf][4 * j + 1], acc[m][nf][4 * j + 2], acc[m][nf][4 * j + 3]}; \
          if (F16) v = v * winv; /* the fp16 weight image carries a power-of-two scale (w16_scale) */ \
          if (FWD) v = v + b_;                                                                       \
          if (EPI) {                                                                                 \
            if (a.status && yo_[m] >= 0 && !(isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2]) && isfinite(v[3]))) \
              atomicOr(a.status, 1);                                                                 \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                         \
              v[i_] = fmaxf(fmaf(v[i_], esc_[i_], esh_[i_]), 0.f) * emk_[i_];                        \
          }                                                                                          \
          if (yo_[m] >= 0) {                                                                         \
            if (c_ < a.cout_store) {                                                                 \
              if (S16) {                                                                             \
                et4 o_;                                                                              \
                o_[0] = (ET)v[0];                                                                    \
                o_[1] = (ET)v[1];                                                                    \
                o_[2] = (ET)v[2];                                                                    \
                o_[3] = (ET)v[3];                                                                    \
                *reinterpret_cast<et4*>(yimg_ + yo_[m] + nf * 32 + 8 * j) = o_;                      \
              } else {                                                                               \
                *reinterpret_cast<f32x4*>(yimg_ + yo_[m] + nf * 32 + 8 * j) = v;                     \
              }                                                                                      \
            }                                                                                        \
            if (STATS) {                                                                             \
              t4_ += v;                                                                              \
              q4_ += v * v;                                                                          \
            }                                                                                        \
          }                                                                                          \
        }                                                                                            \
        if (STATS) { /* the lane keeps channels 2 odd .. + 1 of the quad and gets its neighbour's share of them */ \
          const float k0_ = odd ? t4_[2] : t4_[0], k1_ = odd ? t4_[3] : t4_[1];                      \
          const float g0_ = odd ? t4_[0] : t4_[2], g1_ = odd ? t4_[1] : t4_[3];                      \
          s1[nf][j] += f32x2{k0_ + dpp_xor1(g0_), k1_ + dpp_xor1(g1_)};                              \
          const float l0_ = odd ? q4_[2] : q4_[0], l1_ = odd ? q4_[3] : q4_[1];                      \
          const float h0_ = odd ? q4_[0] : q4_[2], h1_ = odd ? q4_[1] : q4_[3];                      \
          s2[nf][j] += f32x2{l0_ + dpp_xor1(h0_), l1_ + dpp_xor1(h1_)};                              \
        }                                                                                            \
      }                                                                                              \
    _Pragma("unroll") for (int m = 0; m < MF; ++m)                                                   \
      _Pragma("unroll") for (int nf = 0; nf < NF; ++nf)                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) acc[m][nf][i_] = 0.f;                      \
  }
  // ---- deferred epilogue (DEFER) ----
  // tile TI is complete: its accumulators move to `sav`; per fragment the lane's store address (quad 0) and whether its
  // pixel counts.  (The accumulators themselves are not cleared: the next tile's first MFMAs take a zero C operand.)
#define WC_SNAPSHOT(TI)                                                                              \
  {                                                                                                  \
    int t_ = vbx + (TI) * gx;                                                                        \
    const int tx_ = t_ % tilesX;                                                                     \
    t_ /= tilesX;                                                                                    \
    const int ty_ = t_ % tilesY;                                                                     \
    const int n_ = t_ / tilesY;                                                                      \
    const int y0_ = ty_ * TR, x0_ = tx_ * TC;                                                        \
    sv_rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<OT_*>(a.y) + (size_t)n_ * a.Ho * a.Wo * a.ldy, 0, \
                                                a.Ho * a.Wo * a.ldy * (int)sizeof(OT_), 0x00020000); \
    _Pragma("unroll") for (int m = 0; m < MF; ++m) {                                                 \
      const int oy = y0_ + (prc[m] >> 16), ox = x0_ + (prc[m] & 0xffff);                             \
      sv_off[m] = (oy < a.Ho && ox < a.Wo) ? (unsigned)(((oy * a.Wo + ox) * a.ldy + co0 + 4 * h) * (int)sizeof(OT_)) : kNoStore; \
      sav[m] = acc[m][0];                                                                            \
    }                                                                                                \
  }
  // piece K = (channel quad j = K / 4, fragment m = K % 4) of the saved tile: scale / bias, one 16-byte store, the
  // BatchNorm sums in WC_EPILOGUE's order (over m for a quad, then the DPP reduce-scatter step)
#define WC_PIECE(K)                                                                                  \
  {                                                                                                  \
    const int j_ = (K) >> 2, m_ = (K) & 3;                                                           \
    if (FWD) ep_b = *reinterpret_cast<const f32x4*>(epi + 8 * j_ + 4 * h);                           \
    if (m_ == 0) {                                                                                   \
      ep_t4 = f32x4{0.f, 0.f, 0.f, 0.f};                                                             \
      ep_q4 = ep_t4;                                                                                 \
    }                                                                                                \
    f32x4 v = f32x4{sav[m_][4 * j_], sav[m_][4 * j_ + 1], sav[m_][4 * j_ + 2], sav[m_][4 * j_ + 3]};  \
    if (F16) v = v * winv;                                                                           \
    if (FWD) v = v + ep_b;                                                                           \
    /* a padded channel quad is not stored (co0 + 8 j >= cout_store: wave-uniform — sched::wide_config keeps launches whose  \
       stored channels are not a multiple of 8, the 4-channel image gradient, off this kernel) */    \
    const unsigned vo_ = co0 + 8 * j_ < a.cout_store ? sv_off[m_] : kNoStore;                        \
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), sv_rsrc, (int)vo_, 8 * j_ * (int)sizeof(OT_), 0); \
    if (STATS) {                                                                                     \
      const bool in_ = sv_off[m_] != kNoStore;                                                       \
      const f32x4 w_ = f32x4{in_ ? v[0] : 0.f, in_ ? v[1] : 0.f, in_ ? v[2] : 0.f, in_ ? v[3] : 0.f}; \
      ep_t4 += w_;                                                                                   \
      ep_q4 += w_ * w_;                                                                              \
      if (m_ == MF - 1) {                                                                            \
        const float k0_ = odd ? ep_t4[2] : ep_t4[0], k1_ = odd ? ep_t4[3] : ep_t4[1];                \
        const float g0_ = odd ? ep_t4[0] : ep_t4[2], g1_ = odd ? ep_t4[1] : ep_t4[3];                \
        lsum[(2 * j_) * 256] += f32x2{k0_ + dpp_xor1(g0_), k1_ + dpp_xor1(g1_)};                     \
        const float l0_ = odd ? ep_q4[2] : ep_q4[0], l1_ = odd ? ep_q4[3] : ep_q4[1];                \
        const float h0_ = odd ? ep_q4[0] : ep_q4[2], h1_ = odd ? ep_q4[1] : ep_q4[3];                \
        lsum[(2 * j_ + 1) * 256] += f32x2{l0_ + dpp_xor1(h0_), l1_ + dpp_xor1(h1_)};                 \
      }                                                                                              \
    }                                                                                                \
  }
  // a slot of the first phase of a tile that also carries a piece: its fragment reads (+ the piece's bias read), then
  // its MFMAs with the piece's vector instructions between them, the store last
#define WC_PIN_P(NREADS)                                                                             \
  __builtin_amdgcn_sched_group_barrier(0x100, (NREADS), 0);                                          \
  _Pragma("unroll") for (int q_ = 0; q_ < 3 * SM * NF; ++q_) {                                       \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                               \
    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);                                               \
  }                                                                                                  \
  __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
  // the slots of one phase; PIECES: the phase is the first of a tile and carries the deferred epilogue of the previous one
#define WC_BODY(P, FIRST, PIECES)                                                                    \
  _Pragma("unroll") for (int t = 0; t < TPP; ++t)                                                    \
    _Pragma("unroll") for (int s = 0; s < NSLOT; ++s) {                                              \
      const int g_ = t * NSLOT + s;                                                                  \
      const int ws_ = (t & 1) ? (P) : ((P) ^ 1);                                                     \
      if (!(t == TPP - 1 && s == NSLOT - 1)) {                                                       \
        if (s + 1 < NSLOT) {                                                                         \
          WC_READ_A((g_ + 1) & 1, t, s + 1)                                                          \
        } else {                                                                                     \
          WC_READ_W(ws_ ^ 1, t + 1)                                                                  \
          WC_READ_A((g_ + 1) & 1, t + 1, 0)                                                          \
        }                                                                                            \
        WC_MFMA_Z(g_ & 1, ws_, s, (PIECES) && t == 0)                                                \
        const bool piece_ = (PIECES) && g_ < 4 * MF;                                                 \
        if (piece_) {                                                                                \
          WC_PIECE(g_)                                                                               \
        }                                                                                            \
        /* (sched_group_barrier takes literal counts: one call per case) */                          \
        if (piece_) { /* never in the FIRST phase */                                                 \
          if (s + 1 < NSLOT) {                                                                       \
            WC_PIN_P(2 * SM + (FWD ? 1 : 0))                                                         \
          } else {                                                                                   \
            WC_PIN_P(2 * NF + 2 * SM + (FWD ? 1 : 0))                                                \
          }                                                                                          \
        } else if ((FIRST) && g_ == 0) { /* the phase's opening reads sit in front of this slot too */ \
          WC_PIN(2 * NF + 2 * SM + 2 * SM)                                                           \
        } else if (s + 1 < NSLOT) {                                                                  \
          WC_PIN(2 * SM)                                                                             \
        } else {                                                                                     \
          WC_PIN(2 * NF + 2 * SM)                                                                    \
        }                                                                                            \
      }                                                                                              \
    }
  // One phase.  On entry the previous phase's last slot is pending (pixel set 1, weight set P): it is multiplied
  // behind the barrier, under the first reads of this phase.  Slots alternate the pixel sets (an even number per
  // phase), taps alternate the weight sets (an odd number per phase, so consecutive phases swap P).
#define WC_PHASE(P, FIRST)                                                                           \
  {                                                                                                  \
    const int j_ = ph / PARTS, r_ = ph - PARTS * j_;                                                 \
    const unsigned char* xb_ = xs + (j_ & 1) * kXBytes;                                              \
    const unsigned char* wb_ = ws + wslot * WPHB;                                                    \
    const int ro_ = r_ * row1;                                                                       \
    (void)ro_;                                                                                       \
    __syncthreads(); /* this phase is staged; the buffers of the previous one are released */        \
    WC_READ_W((P) ^ 1, 0)                                                                            \
    WC_READ_A(0, 0, 0)                                                                               \
    bool pieces_ = false;                                                                            \
    if (!(FIRST)) {                                                                                  \
      WC_MFMA(1, P, NSLOT - 1)                                                                       \
      WC_PIN(2 * NF + 2 * SM)                                                                        \
      if (tile_done) {                                                                               \
        if (DEFER) {                                                                                 \
          WC_SNAPSHOT(ti)                                                                            \
          pieces_ = true;                                                                            \
        } else {                                                                                     \
          WC_EPILOGUE(ti)                                                                            \
        }                                                                                            \
        ++ti;                                                                                        \
      }                                                                                              \
    }                                                                                                \
    if (DEFER && pieces_) {                                                                          \
      WC_BODY(P, FIRST, true)                                                                        \
    } else {                                                                                         \
      WC_BODY(P, FIRST, false)                                                                       \
    }                                                                                                \
    tile_done = r_ == PARTS - 1 && (j_ + 1) % nchunks == 0;                                          \
    ++ph;                                                                                            \
    wslot = wslot + 1 == NWB ? 0 : wslot + 1;                                                        \
  }
  // the last tile: nothing follows that its epilogue could hide under
#define WC_FINAL_EPILOGUE(TI)                                                                        \
  if (DEFER) {                                                                                       \
    WC_SNAPSHOT(TI)                                                                                  \
    _Pragma("unroll") for (int k_ = 0; k_ < 4 * MF; ++k_) {                                          \
      WC_PIECE(k_)                                                                                   \
    }                                                                                                \
  } else {                                                                                           \
    WC_EPILOGUE(TI)                                                                                  \
  }

  int ph = 0, ti = 0, wslot = 0;
  bool tile_done = false;
  if (nphases > 0) {
    WC_PHASE(1, true)  // leaves weight set 0 pending
    while (ph + 1 < nphases) {
      WC_PHASE(0, false)
      WC_PHASE(1, false)
    }
    if (ph < nphases) {
      WC_PHASE(0, false)
      WC_MFMA(1, 1, NSLOT - 1)
    } else {
      WC_MFMA(1, 0, NSLOT - 1)
    }
    WC_FINAL_EPILOGUE(ti)
  }
  if (SUMB) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s1[0][j] = lsum[(2 * j) * 256];
      s2[0][j] = lsum[(2 * j + 1) * 256];
    }
  }
  __syncthreads();  // matches the producers' last barrier
  if (STATS) {
    // one partial-statistics row per workgroup: lanes of equal (px & 1, h) hold the same channels
    float* red = reinterpret_cast<float*>(xs);  // [4 waves][2][NB]
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int d = 2; d < 32; d <<= 1) {
            s1[nf][j][i] += __shfl_xor(s1[nf][j][i], d);
            s2[nf][j][i] += __shfl_xor(s2[nf][j][i], d);
          }
        if (px < 2) {
          const int c = nf * 32 + 8 * j + 4 * h + 2 * px;
          *reinterpret_cast<f32x2*>(red + (wave * 2 + 0) * NB + c) = s1[nf][j];
          *reinterpret_cast<f32x2*>(red + (wave * 2 + 1) * NB + c) = s2[nf][j];
        }
      }
    __syncthreads();  // also executed by the producers
    if (tid < 2 * NB) {
      const int which = tid / NB, c = tid - which * NB;
      const float v = (red[(0 * 2 + which) * NB + c] + red[(1 * 2 + which) * NB + c]) +
                      (red[(2 * 2 + which) * NB + c] + red[(3 * 2 + which) * NB + c]);
      // (a.stats == nullptr: a forward that needs neither statistics nor the inference epilogue — eval mode with autograd)
      if (a.stats && co0 + c < a.cout_pad) a.stats[((size_t)vbx * 2 + which) * a.cout_pad + co0 + c] = v;
    }
  }
#undef WC_READ_W
#undef WC_READ_A
#undef WC_MFMA
#undef WC_EPILOGUE
#undef WC_FINAL_EPILOGUE
#undef WC_SNAPSHOT
#undef WC_PIECE
#undef WC_PIN_P
#undef WC_BODY
#undef WC_MFMA_Z
#undef WC_PIN
#undef WC_PHASE
}

// ---------------------------------------------------------------------------------------------------------------
static int wide_force() {
  static const int f = [] {
    const char* e = getenv("MIMO_CONV_WIDE");  // 0: never; 2: whenever the geometry is supported; default: cost rule
    const int v = e ? atoi(e) : 1;
    return v == 0 ? -1 : v >= 2 ? 1 : 0;
  }();
  return f;
}

// 0 = the layer runs on conv_bf16x3.hip; else the packed weight rows of the wide layout (ConvLaunch::wide)
int conv3x3_wide_rows(int mode, int N, int cin_p, int rows, int Ho, int Wo) {
  return sched::wide_config(mode, N, cin_p, rows, Ho, Wo, wide_force()).rows_pad;
}
// 16-bit elements of the packed wide weight image
size_t conv3x3_wide_weight_elems(int cin_p, int rows_pad) { return (size_t)ceil_div(cin_p, 16) * 9 * rows_pad * 32; }
int conv3x3_wide_stat_rows() { return 256; }  // one row per persistent workgroup column

int conv3x3_wide_launch(const ConvLaunch& a, int mode, int* rows, hipStream_t stream) {
  const sched::WideCfg c = sched::wide_config(mode, a.N, a.cin_p, a.cout_store, a.Ho, a.Wo, 1);
  if (!a.wpk || c.nf == 0 || c.rows_pad != a.wide || a.ldx % 4 != 0 || a.ldy % 4 != 0 || a.Hi < 2 || a.Wi < 2) {
    set_error("conv3x3 wide: weights packed for %d rows, launch geometry gives %d (nf %d)", a.wide, c.rows_pad, c.nf);
    return MIMO_ERR_INVALID;
  }
  if (mode == 0 && a.cout_store % 8 != 0) {
    set_error("conv3x3 wide data gradient: %d stored channels (a multiple of 8 is needed)", a.cout_store);
    return MIMO_ERR_INVALID;
  }
  if ((int64_t)a.Ho * a.Wo * a.ldy * 4 >= (int64_t)0x7fff0000) {  // 32-bit store offsets of the deferred epilogue (kNoStore)
    set_error("conv3x3 wide: output image of %d x %d x %d exceeds the kernel's 32-bit offsets", a.Ho, a.Wo, a.ldy);
    return MIMO_ERR_INVALID;
  }
  if ((int64_t)a.Hi * a.Wi * a.ldx * 4 > (int64_t)INT32_MAX) {  // 32-bit per-unit source offsets (WD_OFFS)
    set_error("conv3x3 wide: input image of %d x %d x %d exceeds the kernel's 32-bit offsets", a.Hi, a.Wi, a.ldx);
    return MIMO_ERR_INVALID;
  }
  const int tilesY = ceil_div(a.Ho, c.TR), tilesX = ceil_div(a.Wo, c.TC);
  const int numTiles = a.N * tilesY * tilesX;
  const int coTiles = c.rows_pad / (c.nf * 32);
  const int gx = sched::wide_grid_x(numTiles, coTiles);
  if (rows) *rows = gx;
  dim3 grid(gx * coTiles);
#define WIDE_LAUNCH(NF_, MODE_, TPP_, EPI_)                                                                        \
  hipLaunchKernelGGL((conv3x3_wide_kernel<NF_, MODE_, TPP_, EPI_>), grid, dim3(512), 0, stream, a, c.TR, c.TC, tilesY, \
                     tilesX, numTiles, gx, coTiles)
  const bool fwd = mode == 1 || mode == 4 || mode == 6;
  if (fwd && (!a.bias || (a.ep_scale && (a.stats || !a.ep_shift)))) {
    set_error("conv3x3 wide forward: needs the bias; statistics rows and the inference epilogue exclude each other");
    return MIMO_ERR_INVALID;
  }
  if (mode >= 4 && (a.ldx % 8 != 0 || a.cin_p % 8 != 0)) {
    set_error("conv3x3 wide, 16-bit storage: channel counts must be multiples of 8");
    return MIMO_ERR_INVALID;
  }
  if (a.in_scale && (mode != 1 || a.ep_scale || !a.in_shift)) {
    set_error("conv3x3 wide: the input BatchNorm + ReLU can be fused into the split16 training forward only");
    return MIMO_ERR_INVALID;
  }
  if (a.in_scale) {
    if (c.nf == 2)
      hipLaunchKernelGGL((conv3x3_wide_kernel<2, 1, 3, false, true>), grid, dim3(512), 0, stream, a, c.TR, c.TC, tilesY, tilesX,
                         numTiles, gx, coTiles);
    else
      hipLaunchKernelGGL((conv3x3_wide_kernel<1, 1, 9, false, true>), grid, dim3(512), 0, stream, a, c.TR, c.TC, tilesY, tilesX,
                         numTiles, gx, coTiles);
    MIMO_KERNEL_CHECK();
    return MIMO_OK;
  }
#define WIDE_FWD(MODE_)            \
  if (a.ep_scale) {                \
    if (c.nf == 2)                 \
      WIDE_LAUNCH(2, MODE_, 3, true);  \
    else                           \
      WIDE_LAUNCH(1, MODE_, 9, true);  \
  } else {                         \
    if (c.nf == 2)                 \
      WIDE_LAUNCH(2, MODE_, 3, false); \
    else                           \
      WIDE_LAUNCH(1, MODE_, 9, false); \
  }
#define WIDE_DG(MODE_)             \
  if (c.nf == 2)                   \
    WIDE_LAUNCH(2, MODE_, 3, false);   \
  else                             \
    WIDE_LAUNCH(1, MODE_, 9, false);
  switch (mode) {
    case 1: WIDE_FWD(1) break;
    case 4: WIDE_FWD(4) break;
    case 6: WIDE_FWD(6) break;
    case 0: WIDE_DG(0) break;
    case 5: WIDE_DG(5) break;
    default: WIDE_DG(7) break;
  }
#undef WIDE_FWD
#undef WIDE_DG
#undef WIDE_LAUNCH
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

// weight packing: torch OIHW -> [chunk][tap][rows_pad][64 bytes]: split16 modes: chunks of 16 input channels, row =
// [hi 16 | lo 16]; 16-bit storage modes (single): chunks of 32 input channels, row = 32 values.  fp16 images carry
// x 2^8.  Element formulas as pack_weights_bf16x3_kernel (conv_bf16x3.hip): row / column maps, transposed = data gradient.
template <bool F16, bool SINGLE>
__global__ void pack_weights_wide_kernel(const float* __restrict__ w, void* __restrict__ dstv, int cout, int cin,
                                         int rows_pad, int cols, int nchunks, const int* __restrict__ row_map,
                                         const int* __restrict__ col_map, int nrows_map, int transposed) {
  typedef typename std::conditional<F16, _Float16, __bf16>::type ET;
  constexpr int CK = SINGLE ? 32 : 16;
  ET* dst = reinterpret_cast<ET*>(dstv);
  const int total = nchunks * 9 * rows_pad * CK;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int k = i % CK;
    int rest = i / CK;
    const int row = rest % rows_pad;
    rest /= rows_pad;
    const int tap = rest % 9, chunk = rest / 9;
    const int col = chunk * CK + k;
    float v = 0.f;
    if (col < cols && row < nrows_map) {
      const int rm = row_map[row], cm = col_map[col];
      if (rm >= 0 && cm >= 0) {
        const int co = transposed ? cm : rm, ci = transposed ? rm : cm;
        const int kh = transposed ? 2 - tap / 3 : tap / 3, kw = transposed ? 2 - tap % 3 : tap % 3;
        v = w[(((size_t)co * cin + ci) * 3 + kh) * 3 + kw];
      }
    }
    if (F16) v *= kWideF16Scale;
    const ET hi = (ET)v;
    ET* d = dst + (((size_t)chunk * 9 + tap) * rows_pad + row) * 32;
    d[k] = hi;
    if (!SINGLE) d[16 + k] = (ET)(v - (float)hi);
  }
}

// single != 0: the 16-bit storage modes' image (one value per weight, 32-channel chunks)
int pack_weights_wide_launch(const float* w, void* dst, int f16, int cout, int cin, int rows_pad, int cols,
                             const int* row_map, const int* col_map, int nrows_map, int transposed, hipStream_t stream,
                             int single) {
  const int nchunks = ceil_div(cols, single ? 32 : 16);
  const int total = nchunks * 9 * rows_pad * (single ? 32 : 16);
  const int blocks = min(ceil_div(total, 256), 4096);
#define PW_LAUNCH(F_, S_)                                                                                              \
  hipLaunchKernelGGL((pack_weights_wide_kernel<F_, S_>), dim3(blocks), dim3(256), 0, stream, w, dst, cout, cin, rows_pad, \
                     cols, nchunks, row_map, col_map, nrows_map, transposed)
  if (f16) {
    if (single)
      PW_LAUNCH(true, true);
    else
      PW_LAUNCH(true, false);
  } else {
    if (single)
      PW_LAUNCH(false, true);
    else
      PW_LAUNCH(false, false);
  }
#undef PW_LAUNCH
  MIMO_KERNEL_CHECK();
  return MIMO_OK;
}

}  // namespace mimo
