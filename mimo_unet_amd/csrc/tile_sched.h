// Host-side schedulers of libmimo_hip.so: tile shapes, channel-tile widths, split counts, the XCD workgroup order.
// Pure integer arithmetic with no HIP dependency, so that tests/host/sched_test.cpp compiles this header with
// g++ -fsanitize=address,undefined and sweeps every function over its whole argument range (tests/test_sched_cpu.py).
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define MIMO_SCHED_HD __host__ __device__
#else
#define MIMO_SCHED_HD
#endif

namespace mimo {
namespace sched {

MIMO_SCHED_HD static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
MIMO_SCHED_HD static inline int rup(int a, int b) { return cdiv(a, b) * b; }

// Workgroups are dealt round-robin to the 8 XCDs in linear-id order; this maps a linear id to a virtual index such
// that every XCD owns one CONTIGUOUS range of virtual indices.  Bijective on [0, total) for any total >= 1.
// (the kernels call this same function: common.h xcd_virtual_index)
MIMO_SCHED_HD static inline int xcd_virtual_index(int linear, int total) {
  const int k = linear & 7, slot = linear >> 3;
  const int base = total >> 3, rem = total & 7;
  return k * base + (k < rem ? k : rem) + slot;
}

// Output tile TR x TC of `npix` pixels whose halo (TR + 2) x (TC + 2) fits `maxpix` LDS rows: the shape that wastes
// the fewest tile slots over an Ho x Wo image, ties broken by the smaller halo.  TR, TC >= 1; TR * TC <= npix;
// (TR + 2) * (TC + 2) <= maxpix whenever some shape satisfies it (callers pass maxpix >= 3 * (4 + 2)).
static inline void pick_tile_n(int Ho, int Wo, int npix, int maxpix, int* TR, int* TC) {
  double best_eff = -1.0;
  int best_tr = 1, best_tc = 4, best_pix = 1 << 30;
  for (int k = 1; k <= Wo; ++k) {
    int tc = cdiv(Wo, k);
    if (tc > npix) continue;
    int tr = npix / tc;
    if (tr > Ho) tr = Ho;
    while (tr > 1 && (tr + 2) * (tc + 2) > maxpix) --tr;
    if (tr < 1 || (tr + 2) * (tc + 2) > maxpix) continue;
    double eff = double(Ho) * Wo / (double(cdiv(Ho, tr)) * cdiv(Wo, tc) * npix);
    int pix = (tr + 2) * (tc + 2);
    if (eff > best_eff + 1e-9 || (eff > best_eff - 1e-9 && pix < best_pix)) {
      best_eff = eff;
      best_tr = tr;
      best_tc = tc;
      best_pix = pix;
    }
    if (tc <= 4) break;
  }
  *TR = best_tr;
  *TC = best_tc;
}

// ---- 256-pixel convolutions (conv3x3.hip fp32-MFMA kernel, conv_bf16x3.hip) ---------------------------------------
// fragments of 16 output channels per workgroup: the width in 2..4 that pads the channel count least (ties -> wider)
static inline int conv_pick_nfrag(int cout) {
  const int nfr = cdiv(cout, 16);
  if (nfr <= 1) return 1;
  int best = 2, best_cost = 1 << 30;
  for (int nf = 4; nf >= 2; --nf) {
    const int cost = rup(nfr, nf);
    if (cost < best_cost) {
      best_cost = cost;
      best = nf;
    }
  }
  return best;
}
static inline int conv_cout_pad(int cout) { return rup(cdiv(cout, 16), conv_pick_nfrag(cout)) * 16; }

// ---- split weight gradient (wgrad_split.hip) --------------------------------------------------------------------
constexpr int kWgTC = 32;  // pixel-tile width of the weight-gradient kernels
// channel tile (32 / 48 / 64) that pads least; 48 only where it saves >= 30 % of padded work
static inline int wg_pick_ctile(int c_p) {
  if (c_p <= 32) return 32;
  const int p64 = rup(c_p, 64), p48 = rup(c_p, 48);
  return double(p48) <= 0.70 * p64 ? 48 : 64;
}
static inline void wg_tiles(int cin_p, int cout_p, bool ws_enabled, int* CI, int* CO) {
  *CI = wg_pick_ctile(cin_p);
  *CO = wg_pick_ctile(cout_p);
  // the wave-specialised kernel (64 input channels per workgroup) also comes 32 and 48 output channels wide and keeps
  // its efficiency there: take the 48-wide tile whenever it saves >= 15 % of padded work
  if (*CI == 64 && ws_enabled && cout_p > 32 && rup(cout_p, 48) <= 0.85 * rup(cout_p, 64)) *CO = 48;
  // ... and it comes 32 input channels wide (4-row tiles): slower per padded MFMA (~320 against ~400 TFLOP/s), so only where
  // 32-channel tiles save >= 20 % of padded work — 90 -> 45 at 256x256 (96 instead of 128 padded input channels: weight
  // gradient 577 -> ~500 us, step -0.2 ms in three alternating pairs, round 4), 84-channel inputs of fbc = 21
  if (*CI == 64 && ws_enabled && rup(cin_p, 32) <= 0.80 * rup(cin_p, 64)) *CI = 32;
}
static inline bool wg_use_ws(int CI, int CO, bool ws_enabled) {
  return ws_enabled && (CI == 64 || CI == 32) && (CO == 32 || CO == 48 || CO == 64);
}
// tile rows of the wave-specialised kernel: 2 with 64 input channels of fp32 / pair-record operands (128 KB of LDS),
// 4 with 32 input channels, and 4 when BOTH operands are 16-bit tensors (s16: half the LDS bytes per pixel)
static inline int wg_ws_tr(int CI, bool s16 = false) { return (CI == 32 || s16) ? 4 : 2; }
static inline int wg_num_tiles(int N, int H, int W, int tr) { return N * cdiv(H, tr) * cdiv(W, kWgTC); }
// pixel splits (slabs) of a layer's weight gradient: >= 1, <= min(tiles, 1024)
static inline int wg_pick_splits(int N, int H, int W, int cin_pad, int cout_pad, int CI, int CO, bool ws_enabled, int mode,
                                 bool s16 = false, int cus = 256) {
  const int wtiles = (cin_pad / CI) * (cout_pad / CO);
  const bool ws = wg_use_ws(CI, CO, ws_enabled);
  const int tiles = wg_num_tiles(N, H, W, ws ? wg_ws_tr(CI, s16) : 4);
  if (!ws || mode == 0) {
    int splits = cdiv(512, wtiles);  // one 4-wave workgroup per CU: ~2 rounds of workgroups
    if (splits > tiles) splits = tiles;
    if (splits > 1024) splits = 1024;
    if (splits < 1) splits = 1;
    return splits;
  }
  // wave-specialised kernel: one workgroup per CU (128 KB of LDS), all workgroups of a launch do the same work, so
  // time ~ rounds x (pixel tiles per workgroup + fixed cost); the fixed cost (147 KB slab written per workgroup and
  // re-read by the reduction, pipeline fill) is worth about `kFixed` pixel tiles.
  const int kCUs = cus;  // CUs the launch may fill (256 = the chip)
  const int kFixed = 16 / wg_ws_tr(CI, s16);
  int best = 1;
  long bestCost = -1;
  for (int s = 1; s <= tiles && s <= 1024; ++s) {
    const long rounds = cdiv(wtiles * s, kCUs);
    const long cost = rounds * (cdiv(tiles, s) + kFixed);
    if (bestCost < 0 || cost < bestCost) bestCost = cost, best = s;
  }
  return best;
}

// CUs the weight-gradient launches of a plan are sized for (round 5).  They run on a side stream beside the main stream's
// kernels; both are persistent grids of one workgroup per CU that cannot share a CU (128-140 KB of LDS each), so a
// weight-gradient launch that fills the chip makes the main stream's next kernel wait for whole workgroups.  At 32 images
// that costs nothing (every launch is many rounds long); at a few images per GPU — the per-GPU batches of a strong-scaling
// run — the main stream's kernels are 64-256 workgroups of one or two tiles, and leaving them part of the chip is worth
// 5 % of the step at 4 images (4.66 -> 4.40 ms with 128 CUs; 160: 4.43, 112: 4.50, 96: 4.54, 64: 4.93), 2 % at 8 and 16
// (192 CUs: 7.52 -> 7.36, 13.0 -> 12.8 ms; 128: 7.42, 12.85), nothing at 32 (23.90 vs 23.92) — profiles/r05/exp/wgrad_cu_share.txt.
// The size of a plan's launches grows with the pixels of its input AND with the width of the network, so the rule is on
// work = images x height x width x (subnetworks x filter_base_count), in units of cfg3's (S x fbc = 60) images of 256 x 256:
// cfg4 (S = 4: twice the width) at 16 images behaves like cfg3 at 32 — 0.5 % slower with 192 CUs, 3 % slower with 128.
static inline int wg_side_cus(long pixels, int width) {
  const long work = pixels * (long)width, unit = 65536L * 60;
  return work <= 6 * unit ? 128 : work <= 24 * unit ? 192 : 256;
}

// ---- power-of-two scales of the fp16 operands (round 5) ---------------------------------------------------------------
// Float bits in, a power of two out: pure integer work on the exponent field, exact by construction.
MIMO_SCHED_HD static inline float pow2_bits(int k) {  // 2^k for -126 <= k <= 127
  union { unsigned u; float v; } c;
  c.u = (unsigned)(127 + k) << 23;
  return c.v;
}
// Scale of the fp16 (hi, lo) weight images of the split16 / 16-mixed forward: 2^8 (rounds 1-4: keeps ordinary weights out of
// fp16's subnormals) while the layer's largest |w| is below 128; from there the power of two that puts max |w| into
// [2^13, 2^14) — a layer with |w| >= 256 no longer overflows its hi part (the reference's fp32 Conv2d is finite for any
// fp32 weight, components.py:23,26).  `wmax_bits` = float bits of max |w|; inverse: the reciprocal, applied in the
// convolution's epilogue.
MIMO_SCHED_HD static inline float w16_scale(unsigned wmax_bits, bool inverse) {
  const int e = (int)((wmax_bits >> 23) & 0xffu) - 127;  // floor(log2 max |w|)
  int k = e < 7 ? 8 : 13 - e;                            // scale = 2^k
  k = k < -100 ? -100 : k;
  return pow2_bits(inverse ? -k : k);
}
// Scale of dz in the two-MFMA weight gradient: 2^(14 - floor(log2 max|dz|)), so that the largest scaled |dz| lies in
// [2^14, 2^15) (fp16's largest finite value is 65504); inverse: its reciprocal.  An all-zero tensor scales by 2^125.
MIMO_SCHED_HD static inline float wg_dz_scale(unsigned absmax_bits, bool inverse) {
  int e = (int)((absmax_bits >> 23) & 0xffu);  // biased exponent
  e = e < 16 ? 16 : e > 254 ? 254 : e;
  const int k = 141 - e;  // 14 - (e - 127)
  return pow2_bits(inverse ? -k : k);
}

// ---- wide convolution (conv_wide.hip): 512-pixel tiles, 16-channel K chunks, 32x32x16 MFMA ----------------------
constexpr int kWideNPix = 512;    // output pixels per tile (4 consumer waves x 4 fragments of 32 pixels)
constexpr int kWideMaxPix = 640;  // LDS rows of an input halo tile

struct WideCfg {
  int nf;        // 32-channel tiles per workgroup (1 or 2); 0 = layer stays on the 256-pixel kernel
  int rows_pad;  // packed weight rows = channel tiles x 32 x nf
  int TR, TC;
};

static inline int wide_grid_x(int numTiles, int cotiles);

// Which decomposition a split16 3x3 convolution runs on.  `rows` = output channels of the launch (forward: Cout;
// data gradient: the layer's padded input channels), Ho x Wo its output domain.
// force: -1 = never, 0 = by the cost rule, 1 = whenever the geometry is supported.
static inline WideCfg wide_config(int mode, int N, int cin_p, int rows, int Ho, int Wo, int force) {
  WideCfg c{0, 0, 0, 0};
  const bool s16 = mode >= 4 && mode <= 7;  // 16-bit storage modes: 32-channel chunks, one MFMA per product
  if (force < 0 || !(mode == 0 || mode == 1 || s16) || rows < 1 || cin_p < 16 || Ho < 2 || Wo < 2) return c;
  if (s16 && cin_p % 8 != 0) return c;
  // the split16 data gradient's 32-channel instance stores whole pairs of channel quads (its deferred epilogue tests
  // co0 + 8 j < stored channels per wave, not per lane): the 4-channel image gradient stays on the 256-pixel kernel
  if (mode == 0 && rows % 8 != 0) return c;
  // the kernel carries per-unit source offsets (iy * Wi + ix) * ld * element size as 32-bit integers: a layer whose
  // input image (padded-domain gradients: + 2 rows / columns; ld >= cin_p) reaches 2 GiB stays on the 256-pixel kernel
  // (pixel pitch taken as up to twice the wider channel count: a channel slice of a concat buffer; the launch checks
  // its real geometry and refuses instead of reading out of bounds)
  if (int64_t(Ho + 2) * int64_t(Wo + 2) * int64_t(cin_p < rows ? rows : cin_p) * 8 > int64_t(INT32_MAX)) return c;
  int TR, TC;
  pick_tile_n(Ho, Wo, kWideNPix, kWideMaxPix, &TR, &TC);
  if ((TR + 2) * (TC + 2) > kWideMaxPix) return c;
  const int tiles = N * cdiv(Ho, TR) * cdiv(Wo, TC);
  const double eff = double(N) * Ho * Wo / (double(tiles) * kWideNPix);
  const int r32 = cdiv(rows, 32);
  // channel-tile width: two 32-channel tiles per workgroup unless the padding that costs exceeds what the second tile's
  // reuse of the staged input is worth (~10 %)
  const int nf = (r32 >= 2 && double(rup(r32, 2)) / r32 <= 1.10 + 1e-9) ? 2 : 1;
  const int cotiles = cdiv(r32, nf);
  if (force == 0) {
    // Cost of both decompositions in padded-MFMA units, constants fitted to per-layer timings of cfg3 at batch 32
    // (profiles/r03/conv_layers_ab.txt; the rule loses 0.3 % against picking the faster kernel per layer):
    //   wide:      K padded to 16, channels to 32 * nf, / tile efficiency, x (1 + a / K chunks): the per-tile epilogue
    //              and pipeline fill, a = 1.2 (64-channel tiles) / 0.2 (32-channel tiles, one barrier per chunk)
    //   256-pixel: K padded to 32 (tap pairing: a <= 16-channel tail costs 2/3 of a chunk), channels to 16, x 1.15
    //              (its matrix pipe is ~15 % less busy)
    const int k16 = rup(cin_p, s16 ? 32 : 16);
    int TRo, TCo;
    pick_tile_n(Ho, Wo, 256, 360, &TRo, &TCo);
    const double eff_o = double(Ho) * Wo / (double(cdiv(Ho, TRo)) * cdiv(Wo, TCo) * 256);
    const int tail = cin_p - 32 * (cdiv(cin_p, 32) - 1);
    const double k_o = 32.0 * (cdiv(cin_p, 32) - 1) + ((tail <= 16 && !s16) ? 64.0 / 3.0 : 32.0);
    double t_w = double(k16) * (cotiles * nf * 32) / eff * (1.0 + (nf == 2 ? 1.2 : 0.2) / (k16 / (s16 ? 32 : 16)));
    // (16-bit storage modes, one MFMA per product: the 256-pixel kernel is bound by its staging there, x 1.45)
    double t_o = (s16 ? 1.45 : 1.15) * k_o * rup(rows, 16) / eff_o;
    // Both kernels are persistent grids of (pixel-tile columns x channel tiles) workgroups that all walk the same number
    // of tiles: a launch whose tile count just exceeds one round leaves CUs without a workgroup (4 images per GPU,
    // 480 -> 240 data gradient at 64x64: 36 wide tiles x 8 channel tiles = 18 x 8 = 144 workgroups of 2 tiles each, 136 us
    // against 104 us on the 256-pixel kernel's 230 workgroups; round 4).  Price both by the CUs they occupy — when that is
    // fewer than three quarters of them: the batch-32 fit above already carries the ordinary few-percent quantisation
    // (with the factor applied always, four batch-32 decisions flipped for a net +0.02 ms).
    {
      const int cu_w = wide_grid_x(tiles, cotiles) * cotiles;
      const int nf_o = conv_pick_nfrag(rows);
      const int cot_o = cdiv(cdiv(rows, 16), nf_o);
      const int tiles_o = N * cdiv(Ho, TRo) * cdiv(Wo, TCo);
      int gx_o = 256 / cot_o > 0 ? 256 / cot_o : 1;
      if (gx_o > tiles_o) gx_o = tiles_o;
      const int cu_o = cdiv(tiles_o, cdiv(tiles_o, gx_o)) * cot_o;
      if (cu_w < 192) t_w *= 256.0 / cu_w;
      if (cu_o < 192) t_o *= 256.0 / cu_o;
    }
    // the wide kernel is persistent with one workgroup per CU and nothing overlaps its pipeline fill: it needs a
    // (pixel tile, channel tile) pair for every CU and >= 8 K chunks of work per workgroup (at 4 images per GPU the
    // 64x64 layers have 128 pairs and 30->30 at 256x256 two 2-chunk tiles per workgroup: measured 4.5 % of the step)
    const int gx = 256 / cotiles < tiles ? (256 / cotiles > 0 ? 256 / cotiles : 1) : tiles;
    if (tiles * cotiles < 256 || cdiv(tiles, gx) * (k16 / (s16 ? 32 : 16)) < 8 || t_w >= t_o) return c;
  }
  c.nf = nf;
  c.rows_pad = cotiles * nf * 32;
  c.TR = TR;
  c.TC = TC;
  return c;
}

// persistent grid of the wide kernel: gx pixel-tile columns x cotiles channel tiles, every workgroup of a launch walks
// the same number of tiles (+-1)
static inline int wide_grid_x(int numTiles, int cotiles) {
  int gx = 256 / cotiles;
  if (gx < 1) gx = 1;
  if (gx > numTiles) gx = numTiles;
  const int per = cdiv(numTiles, gx);
  return cdiv(numTiles, per);
}

}  // namespace sched
}  // namespace mimo
