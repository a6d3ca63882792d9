// Host-only sweep of the schedulers in mimo_unet_amd/csrc/tile_sched.h, built with -fsanitize=address,undefined by
// tests/test_sched_cpu.py (race / memory tooling of the host side).
//   * xcd_virtual_index: a bijection on [0, total) for every grid size 1 .. 4096, every XCD a contiguous range
//   * pick_tile_n: for every H, W in 1 .. 300 and both tile sizes, the tiles cover every pixel exactly once, fit the
//     pixel budget and the LDS halo budget
//   * wide_config / wide_grid_x: geometry invariants for every BASELINE layer at N in {1, 2, 4, 16, 32, 64}
//   * weight-gradient channel tiles / split counts: >= 1, slabs within the plan's scratch formula
//   * conv_cout_pad: covers the channels, multiple of the fragment width
//   * w16_scale / wg_dz_scale (round 5): exact powers of two, reciprocal pairs, scaled maxima inside fp16's range
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../mimo_unet_amd/csrc/tile_sched.h"

using namespace mimo::sched;

static int failures = 0;
#define CHECK(cond, ...)                      \
  do {                                        \
    if (!(cond)) {                            \
      if (failures < 20) {                    \
        std::printf("FAIL %s: ", #cond);      \
        std::printf(__VA_ARGS__);             \
        std::printf("\n");                    \
      }                                       \
      ++failures;                             \
    }                                         \
  } while (0)

static void test_xcd() {
  std::vector<int> seen;
  for (int total = 1; total <= 4096; ++total) {
    seen.assign(total, 0);
    for (int lin = 0; lin < total; ++lin) {
      const int v = xcd_virtual_index(lin, total);
      CHECK(v >= 0 && v < total, "total %d linear %d -> %d", total, lin, v);
      if (v >= 0 && v < total) ++seen[v];
    }
    for (int v = 0; v < total; ++v) CHECK(seen[v] == 1, "total %d: virtual index %d hit %d times", total, v, seen[v]);
    // every XCD (linear % 8) owns one contiguous, increasing range
    for (int k = 0; k < 8 && k < total; ++k) {
      int prev = -1;
      for (int lin = k; lin < total; lin += 8) {
        const int v = xcd_virtual_index(lin, total);
        CHECK(prev < 0 || v == prev + 1, "total %d xcd %d: %d after %d", total, k, v, prev);
        prev = v;
      }
    }
  }
}

static void test_tiles() {
  const int cfgs[3][2] = {{256, 360}, {512, 640}, {128, 180}};
  std::vector<unsigned char> cover;
  for (auto& c : cfgs)
    for (int H = 1; H <= 300; ++H)
      for (int W = 1; W <= 300; ++W) {
        int TR = 0, TC = 0;
        pick_tile_n(H, W, c[0], c[1], &TR, &TC);
        CHECK(TR >= 1 && TC >= 1 && TR * TC <= c[0], "%dx%d npix %d: tile %dx%d", H, W, c[0], TR, TC);
        CHECK((TR + 2) * (TC + 2) <= c[1], "%dx%d: halo %d > %d", H, W, (TR + 2) * (TC + 2), c[1]);
        if (TR < 1 || TC < 1) continue;
        if ((H * 7 + W) % 5 != 0 && !(H <= 40 && W <= 40)) continue;  // full cover check on a fifth of the sizes
        cover.assign((size_t)H * W, 0);
        const int ty = cdiv(H, TR), tx = cdiv(W, TC);
        for (int a = 0; a < ty; ++a)
          for (int b = 0; b < tx; ++b)
            for (int i = 0; i < TR * TC; ++i) {
              const int y = a * TR + i / TC, x = b * TC + i % TC;
              if (y < H && x < W) ++cover[(size_t)y * W + x];
            }
        for (size_t i = 0; i < cover.size(); ++i)
          if (cover[i] != 1) {
            CHECK(false, "%dx%d tile %dx%d: pixel %zu covered %d times", H, W, TR, TC, i, cover[i]);
            break;
          }
      }
}

struct Layer {
  int cin, cout, h;
};
static std::vector<Layer> baseline_layers() {
  std::vector<Layer> v;
  // BASELINE configs 2-5 at 256x256: (S, f) = (2, 21), (2, 30), (4, 30), (1, 30); Ci = 3 / 2
  const int sf[4][3] = {{2, 21, 3}, {2, 30, 2}, {4, 30, 2}, {1, 30, 2}};
  for (auto& c : sf) {
    const int S = c[0], f = c[1], Ci = c[2], fs = f * S;
    const Layer l[] = {{Ci, f, 256}, {f, f, 256}, {f, 2 * f, 128}, {2 * f, 2 * f, 128}, {2 * fs, 4 * fs, 64},
                       {4 * fs, 4 * fs, 64}, {4 * fs, 8 * fs, 32}, {8 * fs, 8 * fs, 32}, {8 * fs, 8 * fs, 16},
                       {16 * fs, 8 * fs, 32}, {8 * fs, 4 * fs, 32}, {8 * fs, 4 * fs, 64}, {4 * fs, 2 * fs, 64},
                       {4 * fs, 2 * fs, 128}, {2 * fs, fs, 128}, {f * (S + 1), f * (S + 1) / 2, 256},
                       {f * (S + 1) / 2, f, 256}};
    for (auto& x : l) v.push_back(x);
  }
  return v;
}
static int pad8(int c) { return rup(c, 8); }

static void test_layers() {
  const int batches[6] = {1, 2, 4, 16, 32, 64};
  for (const Layer& L : baseline_layers())
    for (int N : batches) {
      const int cin_p = L.cin <= 4 ? rup(L.cin, 4) : pad8(L.cin), cout_p = pad8(L.cout);
      // --- channel padding of the 256-pixel kernels
      const int cp = conv_cout_pad(L.cout), nf = conv_pick_nfrag(L.cout);
      CHECK(cp >= L.cout && cp % (16 * nf) == 0 && cp < L.cout + 16 * nf, "cout %d: pad %d nfrag %d", L.cout, cp, nf);
      // --- wide convolution: forward (mode 1) and data gradient (mode 0), rule and forced, split16 and 16-bit storage
      const int modes[4] = {1, 0, 4, 5};
      for (int mode : modes)
        for (int force = 0; force <= 1; ++force) {
          const bool fwd = mode == 1 || mode == 4;
          const int K = fwd ? cin_p : cout_p, R = fwd ? cout_p : cin_p, Ho = fwd ? L.h : L.h + 2;
          const WideCfg c = wide_config(mode, N, K, R, Ho, Ho, force);
          if (c.nf == 0) continue;
          CHECK((c.nf == 1 || c.nf == 2) && c.rows_pad >= R && c.rows_pad % (32 * c.nf) == 0 && c.rows_pad < R + 64,
                "wide rows: %d->%d nf %d rows_pad %d", K, R, c.nf, c.rows_pad);
          CHECK(c.TR * c.TC <= kWideNPix && (c.TR + 2) * (c.TC + 2) <= kWideMaxPix, "wide tile %dx%d", c.TR, c.TC);
          const int tiles = N * cdiv(Ho, c.TR) * cdiv(Ho, c.TC), cot = c.rows_pad / (32 * c.nf);
          const int gx = wide_grid_x(tiles, cot);
          CHECK(gx >= 1 && gx <= tiles && (gx * cot <= 256 || gx == 1), "wide grid: tiles %d cot %d gx %d", tiles, cot, gx);
          // every workgroup column walks ceil or floor(tiles / gx) tiles, at least one
          CHECK((tiles - 1 - (gx - 1)) / gx + 1 >= 1, "wide grid: last column without a tile");
          // the launch recomputes the configuration from (mode, geometry) alone with force = 1: same rows
          const WideCfg c2 = wide_config(mode, N, K, R, Ho, Ho, 1);
          CHECK(c2.rows_pad == c.rows_pad && c2.nf == c.nf, "wide: launch sees %d rows, packer %d", c2.rows_pad, c.rows_pad);
        }
      // --- weight gradient
      for (int ws = 0; ws <= 1; ++ws)
        for (int mode = 0; mode <= 3; ++mode) {  // bit 1: both operands 16-bit tensors (4-row tiles)
          const bool s16 = (mode & 2) != 0;
          int CI = 0, CO = 0;
          wg_tiles(cin_p, cout_p, ws != 0, &CI, &CO);
          CHECK((CI == 32 || CI == 48 || CI == 64) && (CO == 32 || CO == 48 || CO == 64), "wg tiles %d %d", CI, CO);
          const int cin_pad = rup(cin_p, CI), cout_pad = rup(cout_p, CO);
          for (const int cus : {256, 192, 128}) {  // launches sized for the whole chip or for a share of it (wg_side_cus)
          const int s = wg_pick_splits(N, L.h, L.h, cin_pad, cout_pad, CI, CO, ws != 0, mode & 1, s16, cus);
          const int tiles = wg_num_tiles(N, L.h, L.h, wg_use_ws(CI, CO, ws != 0) ? wg_ws_tr(CI, s16) : 4);
          CHECK(s >= 1 && s <= 1024 && s <= tiles, "wg splits %d (tiles %d) for %d->%d @%d N %d", s, tiles, L.cin, L.cout, L.h, N);
          // slabs + group-sum levels fit the plan's scratch formula (plan.hip cap_slab: splits + splits / 8 + 2 slabs)
          int extra = 0;
          for (int n = s; n > 1;) {
            n = cdiv(n, 16);
            extra += n;
          }
          CHECK(extra <= s / 8 + 2, "wg reduce levels %d > %d for %d splits", extra, s / 8 + 2, s);
          }
        }
    }
}

// Dispatch decisions the measurements stand on (profiles/r03/conv_layers_ab.txt): a change of the
// cost rule that flips one of these is a deliberate act.  mode 1 = split16 forward, 0 = split16 data gradient
// (rows = padded input channels, domain (H + 2) x (W + 2)), 4 / 5 = bf16-mixed forward / data gradient.
static void test_pinned_decisions() {
  struct D { int mode, N, K, rows, H; bool wide; int nf; };
  const D pins[] = {
      {1, 32, 960, 480, 32, true, 2},   // up1.c1 forward: the largest layer
      {1, 32, 480, 480, 32, true, 2},
      {1, 32, 240, 240, 64, true, 2},
      {1, 32, 32, 30, 256, true, 1},    // 30->30 at 256x256: one 32-channel tile, a chunk per phase
      {1, 32, 48, 30, 256, true, 1},    // 45->30
      {1, 32, 480, 480, 16, false, 0},  // 16x16: the 128-pixel instances fill the chip better
      {1, 32, 96, 45, 256, false, 0},   // 90->45 forward stays on the 256-pixel kernel (64-channel tiles pad 45 -> 64)
      {0, 32, 480, 960, 34, false, 0},  // up1.c1 data gradient: the NF = 4 instance re-reads least
      {0, 32, 32, 32, 258, true, 1},    // 30->30 data gradient
      {1, 4, 32, 30, 256, false, 0},    // 4 images per GPU: two 2-chunk tiles per workgroup do not pay for a persistent kernel
      {1, 4, 240, 240, 64, false, 0},   // ... and 32 pixel tiles x 4 channel tiles leave half the CUs idle
      {0, 32, 240, 480, 66, true, 2},   // up2.c1 data gradient at batch 32: wide
      {0, 4, 240, 480, 66, false, 0},   // ... at 4 images per GPU its 36 x 8 tiles fill 144 workgroups of 2 tiles: the 256-pixel kernel's 230 win (136 -> 104 us)
      {4, 32, 960, 480, 32, true, 2},   // 16-bit storage: the wide kernel takes nearly everything
      {5, 32, 480, 960, 34, true, 2},
  };
  {  // weight-gradient channel tiles (input, output): 32-wide input tiles only where they save >= 20 % of padded work
    struct T { int cin_p, cout_p, CI, CO; };
    const T tiles[] = {{96, 48, 32, 48}, {88, 168, 32, 64}, {48, 32, 64, 32}, {32, 32, 32, 32}, {64, 64, 64, 64},
                       {120, 64, 64, 64}, {480, 480, 64, 64}, {160, 48, 64, 48}};
    for (const T& t : tiles) {
      int CI = 0, CO = 0;
      wg_tiles(t.cin_p, t.cout_p, true, &CI, &CO);
      CHECK(CI == t.CI && CO == t.CO, "pinned weight-gradient tiles for %d x %d: %d x %d (expected %d x %d)", t.cin_p, t.cout_p, CI, CO, t.CI,
            t.CO);
    }
  }
  for (const D& d : pins) {
    const WideCfg c = wide_config(d.mode, d.N, d.K, d.rows, d.H, d.H, 0);
    CHECK((c.nf != 0) == d.wide && (!d.wide || c.nf == d.nf), "pinned decision mode %d N %d %d->%d @%d: nf %d (expected %s nf %d)",
          d.mode, d.N, d.K, d.rows, d.H, c.nf, d.wide ? "wide" : "256-pixel", d.nf);
  }
}

// wide kernel: 32-bit per-unit source offsets - a layer whose image reaches 2 GiB must stay on the 256-pixel kernel
static void test_wide_offset_guard() {
  CHECK(wide_config(1, 1, 32, 32, 1024, 1024, 1).nf != 0, "1024x1024x32 fits 32-bit offsets");
  CHECK(wide_config(1, 1, 128, 128, 2048, 2048, 1).nf == 0, "2048x2048x128 fp32 overflows 32-bit offsets: not wide");
  CHECK(wide_config(1, 1, 32, 32, 4096, 4096, 1).nf == 0, "4096x4096x32 overflows with a 2x pixel pitch: not wide");
  CHECK(wide_config(0, 1, 64, 64, 2048, 2048, 0).nf == 0, "data gradient, same bound");
}

// Power-of-two scales of the fp16 operands (w16_scale, wg_dz_scale): for every exponent and a sweep of mantissas — an exact
// power of two, reciprocal pairs, the scaled maximum inside the band that fp16 represents, and the fixed 2^8 for ordinary weights.
static float bits_to_float(unsigned u) {
  union { unsigned u; float v; } c;
  c.u = u;
  return c.v;
}
static unsigned float_to_bits(float v) {
  union { unsigned u; float v; } c;
  c.v = v;
  return c.u;
}
static void test_scales() {
  for (unsigned e = 1; e <= 254; ++e)
    for (unsigned m = 0; m < (1u << 23); m += 0x1ffffu) {
      const unsigned bits = (e << 23) | m;
      const float x = bits_to_float(bits);
      const float ws = w16_scale(bits, false), wi = w16_scale(bits, true);
      CHECK((float_to_bits(ws) & 0x7fffffu) == 0 && (float_to_bits(wi) & 0x7fffffu) == 0, "w16_scale(%g) not a power of two", x);
      CHECK(ws * wi == 1.0f, "w16_scale(%g): %g * %g != 1", x, ws, wi);
      if (x < 128.f) CHECK(ws == 256.f, "w16_scale(%g) = %g, expected 2^8", x, ws);
      if (x >= 128.f && e <= 127 + 113) CHECK(x * ws >= 8192.f && x * ws < 16384.f, "w16_scale(%g): scaled max %g outside [2^13, 2^14)", x, x * ws);
      CHECK(x * ws < 65504.f || e > 127 + 113, "w16_scale(%g): scaled max %g overflows fp16", x, x * ws);
      const float ds = wg_dz_scale(bits, false), di = wg_dz_scale(bits, true);
      CHECK((float_to_bits(ds) & 0x7fffffu) == 0 && ds * di == 1.0f, "wg_dz_scale(%g): %g, %g", x, ds, di);
      if (e >= 16) CHECK(x * ds >= 16384.f && x * ds < 32768.f, "wg_dz_scale(%g): scaled max %g outside [2^14, 2^15)", x, x * ds);
      if (e < 16) CHECK(x * ds < 32768.f, "wg_dz_scale(%g): scaled max %g", x, x * ds);
    }
  CHECK(wg_dz_scale(0u, false) > 0.f && wg_dz_scale(0u, true) > 0.f, "all-zero dz: finite positive scale");
  CHECK(w16_scale(0u, false) == 256.f, "all-zero weights: 2^8");
  // infinity / NaN maxima: a finite scale (the non-finite values themselves propagate)
  CHECK(wg_dz_scale(0x7f800000u, false) > 0.f && wg_dz_scale(0x7fc00000u, true) > 0.f, "non-finite maximum: finite scale");
}

static void test_side_cus() {
  // the per-GPU batches of cfg3 (S x fbc = 60) under 8-, 4-, 2- and 1-way strong scaling, cfg4 (S = 4: 120) at its 16 images per
  // GPU and cfg2 (2 x 21) at its 64: the measurements the rule stands on (profiles/r05/exp/wgrad_cu_share.txt)
  CHECK(wg_side_cus(4L * 256 * 256, 60) == 128 && wg_side_cus(8L * 256 * 256, 60) == 192 && wg_side_cus(16L * 256 * 256, 60) == 192 &&
            wg_side_cus(32L * 256 * 256, 60) == 256 && wg_side_cus(16L * 256 * 256, 120) == 256 && wg_side_cus(64L * 256 * 256, 42) == 256,
        "wg_side_cus on the measured configurations");
  for (const int width : {8, 42, 60, 120, 512}) {
    int prev = 0;
    for (long px = 1; px < (1L << 34); px = px * 3 / 2 + 1) {  // monotone, 8 <= cus <= 256
      const int c = wg_side_cus(px, width);
      CHECK(c >= 8 && c <= 256 && c >= prev, "wg_side_cus(%ld, %d) = %d", px, width, c);
      prev = c;
    }
  }
}

int main() {
  test_side_cus();
  test_scales();
  test_xcd();
  test_wide_offset_guard();
  test_tiles();
  test_layers();
  test_pinned_decisions();
  if (failures) {
    std::printf("%d check(s) failed\n", failures);
    return 1;
  }
  std::printf("sched_test: all checks passed\n");
  return 0;
}
