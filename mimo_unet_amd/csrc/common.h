// Shared host/device helpers of libmimo_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mimo_hip.h"
#include "tile_sched.h"

namespace mimo {

void set_error(const char* fmt, ...);

#define MIMO_HIP_CHECK(expr)                                                                          \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess) {                                                                           \
      ::mimo::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);   \
      return MIMO_ERR_HIP;                                                                            \
    }                                                                                                 \
  } while (0)
#define MIMO_KERNEL_CHECK() MIMO_HIP_CHECK(hipGetLastError())
#define MIMO_TRY(expr)            \
  do {                            \
    int rc_ = (expr);             \
    if (rc_ != MIMO_OK) return rc_; \
  } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Channel padding of every NHWC activation tensor (floats per pixel).
static inline int pad_channels(int c) { return round_up(c, 8); }

constexpr int kWave = 64;

// Element type of the activation-like tensors in HBM (conv outputs, activations and their gradients): fp32, or bf16 /
// fp16 in the 16-bit storage modes MIMO_PREC_BF16_MIXED / MIMO_PREC_FP16_MIXED.  Arithmetic is always fp32.
enum StoreType { ST_F32 = 0, ST_BF16 = 1, ST_F16 = 2 };
static inline int store_bytes(int dt) { return dt == ST_F32 ? 4 : 2; }

// Workgroups are dealt round-robin to the 8 XCDs (each with a private L2) in linear-id order.  This maps a
// linear workgroup id to a "virtual" index such that every XCD owns one CONTIGUOUS range of virtual indices
// (any total, bijective): kernels then decode the virtual index so that workgroups which read the same
// operand tiles (the channel tiles of one pixel tile; neighbouring pixel tiles sharing halo rows) are
// neighbours in virtual order and therefore meet in one L2 instead of eight.
__device__ __forceinline__ int xcd_virtual_index(int linear, int total) {
  return sched::xcd_virtual_index(linear, total);  // tile_sched.h: host-testable
}

// ------------------------------------------------------------------ conv3x3 (conv3x3.hip)
// "forward-type" 3x3 correlation on the f32 MFMA: y[n,oy,ox,co] = bias[co] +
//   sum_{kh,kw,ci} w[kh,kw][co][ci] * X(n, oy+kh-off, ox+kw-off, ci)
// off == 1: Ho==Hi, X is reflect-padded (forward conv, components.py:23,26)
// off == 2: Ho==Hi+2, X is zero outside the image (transposed conv producing the gradient on
//           the reflect-PADDED domain; consumers fold the border back, see fold_* in elementwise.hip)
// power-of-two scale of a layer's fp16 weight image from the float bits of its max |w| (tile_sched.h: host-testable)
__host__ __device__ __forceinline__ float w16_scale(unsigned wmax_bits, bool inverse) { return sched::w16_scale(wmax_bits, inverse); }

constexpr int kFinSlack = 32;  // zero floats behind ConvLaunch::in_scale / in_shift (a 32-channel chunk may start at cin_p - 8)
struct ConvLaunch {
  const float* x;
  float* y;
  const float* w;     // fp32 kernel: packed [9][cout_pad][cin_p]
  const void* wpk = nullptr;  // split kernel: packed [chunk][tap][cout_pad][hi 32 | lo 32] fp16/bf16
  const float* bias;  // [cout_pad] or nullptr
  float* stats;       // per-block partial sums [blocks][2][cout_pad] or nullptr
  int N, Hi, Wi, ldx, cin_p;
  int Ho, Wo, ldy, cout_pad, cout_store;
  int off;
  // inference epilogue (all null = plain conv output): y = relu(conv * ep_scale[c] + ep_shift[c]) * ep_mask[n][c]
  // — eval-mode BatchNorm + ReLU (+ Dropout2d multipliers, [N][ep_mask_ld]) folded into the store
  const float* ep_scale = nullptr;
  const float* ep_shift = nullptr;
  const float* ep_mask = nullptr;
  int ep_mask_ld = 0;
  int* status = nullptr;  // with ep_scale: bit 0 is OR-ed in when a convolution output is not finite (the ReLU would drop a NaN)
  int pair = 0;  // split kernel: a.wpk holds the tap-paired image of the last chunk (conv3x3_pair_tail)
  // != 0: a.wpk holds the wide kernel's image — [16-channel chunk][tap][wide rows][hi 16 | lo 16] — and the launch runs
  // on conv_wide.hip (value = packed rows, conv3x3_wide_rows)
  int wide = 0;
  // != nullptr (split16 forward on the wave-specialised kernels only, conv3x3_split_fuses_input): x is the PRE-activation
  // tensor z of the producing convolution and the loaders apply its BatchNorm + ReLU on the way into LDS — relu(z *
  // in_scale[c] + in_shift[c]), bn_relu_fwd_kernel's arithmetic (components.py:24-25 between the two convolutions of a
  // DoubleConv) — so that the activated tensor is never written to HBM
  // Both arrays must be readable (and zero) for kFinSlack floats past cin_p: the loaders read whole 16- / 32-channel chunks
  const float* in_scale = nullptr;
  const float* in_shift = nullptr;
  // fp16 weight images (modes 1 and 6): device word with the float bits of the layer's max |w| the image was packed with
  // (w16_scale); nullptr: the image carries the fixed 2^8 (per-operator entry points)
  const unsigned* wmax = nullptr;
  // > 1 (wave-specialised 256-pixel kernel, split16 modes 0 / 1, cin_p a multiple of 32; conv3x3_ksplit): the K walk over
  // the input channels is split ksplit-fold across workgroups; slab k of y — [ksplit][N][Ho][Wo][ldy] — receives the partial
  // sums of chunk range k (bias, stats and the inference epilogue must be null: conv_ksplit_reduce_launch adds them up)
  int ksplit = 1;
};
// K split of a launch by cost (1 = none): few pixel tiles and a long K walk — the 16x16 / 32x32 layers at a few images per
// GPU — finish sooner as wide channel tiles on ksplit x the workgroups plus one reduction pass than as narrow channel tiles
// whose every phase is mostly fixed cost.  MIMO_CONV_KSPLIT: 0 = never, n >= 2 = n wherever the geometry allows (tests).
int conv3x3_ksplit(int mode, int N, int cin_p, int cout_pad, int Ho, int Wo);
// conv3x3_bf16x3_launch with the K split of conv3x3_ksplit when kpart (>= kpart_floats floats of scratch) can hold its
// slabs: the partial launch, then out = bias + sum of the slabs with the BatchNorm partial rows (forward) in one pass
int conv3x3_bf16x3_launch_k(const ConvLaunch& a, int mode, int* rows, hipStream_t stream, float* kpart, size_t kpart_floats);
size_t conv3x3_ksplit_scratch(int mode, int N, int cin_p, int cout_pad, int Ho, int Wo, int ldy);  // floats; 0: no split
// 1 when conv3x3_bf16x3_launch(mode) runs this launch on a kernel whose loaders can apply ConvLaunch::in_scale / in_shift
int conv3x3_split_fuses_input(int mode, int wide, int Ho, int Wo);
// returns number of partial-stat rows (spatial blocks) through *rows when stats != nullptr
int conv3x3_launch(const ConvLaunch& a, int* rows, hipStream_t stream);
// plain-FMA kernels for the image convolution (conv_thin.hip; cin = the logical input channels, 1..4, stored in channels
// [0, cin) of x): the forward on the same ConvLaunch (a.w, the inference epilogue, one BatchNorm partial row per workgroup),
// and the weight gradient from an fp32 dz into torch's OIHW layout (partial: wgrad_thin_scratch floats)
int conv3x3_thin_ok(int cin, int cout_p);
int conv3x3_thin_launch(const ConvLaunch& a, int cin, int* rows, hipStream_t stream);
size_t wgrad_thin_scratch(int cin, int cout_p);
int wgrad_thin_ok(int cin, int cout_p, int N, int H, int W);  // the weight gradient too (few channels, many tiles)
int wgrad_thin_launch(const float* x, int ldx, const float* dz, int lddz, int N, int H, int W, int cin, int cout, int cout_p,
                      float* partial, float* dw, hipStream_t stream);
// split 16-bit variant (conv_bf16x3.hip): same ConvLaunch, reads a.wpk instead of a.w.
// mode 0: split16 data gradient (bf16 pairs, pre-split input); 1: split16 forward (fp16 pairs);
// 2: bf16 forward (one MFMA per product); 3: bf16 data gradient
int conv3x3_bf16x3_launch(const ConvLaunch& a, int mode, int* rows, hipStream_t stream);
int pack_weights_bf16x3_launch(const float* w, void* dst, int f16, int cout, int cin, int rows_pad, int cols,
                               const int* row_map, const int* col_map, int transposed, hipStream_t stream, int pair = 0);
// 1 when conv3x3_bf16x3_launch(mode) runs a layer with cin_p input channels and Ho x Wo outputs on the tap-paired
// instance (last chunk <= 16 channels: two taps per MFMA); the weights must then be packed with pair = 1
int conv3x3_pair_tail(int mode, int cin_p, int Ho, int Wo);
// wide decomposition (conv_wide.hip; which layers: sched::wide_config).  rows = output channels of the launch (forward:
// the padded Cout; data gradient: the layer's padded input channels); returns the packed weight rows, 0 = the layer
// stays on conv_bf16x3.hip.  The packer and the launch must agree (ConvLaunch::wide).
int conv3x3_wide_rows(int mode, int N, int cin_p, int rows, int Ho, int Wo);
size_t conv3x3_wide_weight_elems(int cin_p, int rows_pad);  // 16-bit elements of the packed image
int conv3x3_wide_stat_rows();
int conv3x3_wide_launch(const ConvLaunch& a, int mode, int* rows, hipStream_t stream);
int pack_weights_wide_launch(const float* w, void* dst, int f16, int cout, int cin, int rows_pad, int cols,
                             const int* row_map, const int* col_map, int nrows_map, int transposed, hipStream_t stream,
                             int single = 0);
int conv3x3_pick_nfrag(int cout);            // fragments (of 16 output channels) per workgroup
int conv3x3_cout_pad(int cout);              // packed weight rows for that choice
int conv3x3_stat_rows(int N, int Ho, int Wo);  // spatial workgroups == partial-stat rows
int conv3x3_ws_stat_rows(int N, int Ho, int Wo);  // same for the wave-specialised split kernel (4 rows per tile)

// 256 bytes of zeros in device memory: masked-out tile elements are LOADED from here instead of being
// selected to zero after the load — a select on the loaded value makes the compiler wait for the load
// right where it was issued, which silently removes a register prefetch.
static __device__ __attribute__((aligned(256))) float kZeroPage[64];  // zero-initialised, never written

struct WgradLaunch {
  const float* x;   // [N,H,W,ldx] activations feeding the conv (reflect-padded on the fly)
  const float* dz;  // [N,H,W,lddz]
  float* partial;   // [splits][9][cin_pad][cout_pad]
  int N, H, W, ldx, lddz;
  int cin_p, cout_p;      // valid channel extents in x / dz (multiples of 4)
  int cin_pad, cout_pad;  // multiples of 32
  int splits;
  // split kernels: MFMAs per product block — 3 (bf16 hi/lo pairs both sides), 1 (bf16 compute), or 2 (round 5,
  // wave-specialised kernel, store 0: the activation as one fp16 value, dz as a scaled fp16 pair; needs dz_absmax)
  int np = 3;
  // np == 2: dz_absmax_n floats whose maximum is max |dz| of this tensor — one per workgroup of the launch that wrote dz
  // (bn_bwd_apply / split_pairs: plain stores, no atomics; kDzMaxSlots is their capacity).  Every workgroup of the weight
  // gradient reduces them itself (wg_dz_absmax: identical result everywhere), scales dz by wg_dz_scale(max) and
  // wgrad_reduce_launch divides the result by it again
  const float* dz_absmax = nullptr;
  int dz_absmax_n = 0;
  // split kernels, operand storage: 0 = activations fp32 + dz pre-split bf16 pair records; 1 / 2 = activations and dz
  // plain NHWC bf16 / fp16 (ldx, lddz in elements); 3 / 4 = activations fp32 (the packed image) + dz plain bf16 / fp16
  int store = 0;
  // != nullptr (split16 wave-specialised kernel only, wgrad_split_fuses_input): x is the PRE-activation tensor z of the
  // producing convolution and the loader applies its BatchNorm + ReLU on the way in — a = relu(z * in_scale[c] +
  // in_shift[c]), the arithmetic of bn_relu_fwd_kernel — so that the activated tensor is never materialised
  const float* in_scale = nullptr;
  const float* in_shift = nullptr;
};
// the power of two that scales dz in the two-MFMA weight gradient, from the float bits of max |dz| (tile_sched.h: host-testable)
constexpr int kDzMaxSlots = 2048 * 2;  // workgroups of the largest launch that writes dz (<= 2048 x 2)
__host__ __device__ __forceinline__ float wg_dz_scale(unsigned absmax_bits, bool inverse) { return sched::wg_dz_scale(absmax_bits, inverse); }
#if defined(__HIPCC__)
// max of the n per-workgroup maxima, computed redundantly by every wave that calls it (n <= kDzMaxSlots floats, L2-resident:
// n / 64 coalesced loads per lane and a butterfly) — no hot word that thousands of waves would queue on
__device__ __forceinline__ float wg_dz_absmax(const float* __restrict__ slots, int n) {
  float m = 0.f;
  for (int i = (int)(threadIdx.x & 63); i < n; i += 64) m = fmaxf(m, slots[i]);
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) m = fmaxf(m, __shfl_xor(m, d));
  return m;
}
#endif
int wgrad_launch(const WgradLaunch& a, hipStream_t stream);
int wgrad_split_has_np2(int cin_p, int cout_p);
// 1 when wgrad_split_launch runs this geometry on a kernel that can apply WgradLaunch::in_scale / in_shift in its loader
int wgrad_split_fuses_input(int cin_p, int cout_p, int store, int np);
// split-bf16 variant (wgrad_split.hip): cin_pad / cout_pad must be multiples of its (CI, CO) tile
void wgrad_split_tiles(int cin_p, int cout_p, int* CI, int* CO);
// store: WgradLaunch::store; cus: CUs the launch is sized for (sched::wg_side_cus)
int wgrad_split_pick_splits(int N, int H, int W, int cin_pad, int cout_pad, int CI, int CO, int store = 0, int cus = 256);
int wgrad_split_launch(const WgradLaunch& a, hipStream_t stream);
int wgrad_pick_splits(int N, int H, int W, int cin_pad, int cout_pad);
// extra floats the reduction needs behind the splits*9*cin_pad*cout_pad partial slabs
size_t wgrad_reduce_scratch(int splits, int cin_pad, int cout_pad);
// dW[co][ci][kh][kw] (torch OIHW) = sum over splits of partial[..][tap][cin_map^-1(ci)][co]
// dz_absmax != nullptr: the slabs carry the factor wg_dz_scale(*dz_absmax) (WgradLaunch::np == 2), removed here
int wgrad_reduce_launch(const float* partial, int splits, int cin_pad, int cout_pad, const int* cin_map,
                        int cin_p, int cin, int cout, float* dw, hipStream_t stream, const float* dz_absmax = nullptr,
                        int dz_absmax_n = 0, hipEvent_t done = nullptr);  // done: recorded when the reduction completes

// All weight repacks of a step in ONE launch (a per-layer launch each cost more in dispatch gaps than in
// work): a device table of jobs, blockIdx.y = job.  kind 0: fp32 [tap][rows_pad][cols]; 1 / 2: fp16 / bf16
// (hi, lo) pairs [chunk][tap][rows_pad][hi 32 | lo 32]; 3 / 4: fp16 / bf16 pairs in the wide kernel's layout
// [16-channel chunk][tap][rows_pad][hi 16 | lo 16]; 5 / 6: fp16 (x 2^8) / bf16 single values in the wide kernel's
// 16-bit storage layout [32-channel chunk][tap][rows_pad][32] (kinds 3-6: rows beyond map_rows are zero); bias_n > 0
// additionally copies the layer's bias.
struct PackJob {
  int64_t w_off, bias_off;  // float offsets into the bound parameter buffer
  void* dst;
  float* bias_dst;
  const int *row_map, *col_map;
  int kind, cout, cin, rows_pad, cols, transposed, total, bias_n;
  int pair;  // kind 1 / 2: tap-paired image of the last chunk (conv3x3_pair_tail)
  int map_rows;  // kind 3 / 4: entries of row_map (rows_pad may exceed it)
  unsigned* wmax;  // fp16 kinds (1, 3, 5): max |w| of the layer at this pack, float bits (wabsmax_jobs_launch); image scale = w16_scale
};
int pack_jobs_launch(const PackJob* jobs_dev, int njobs, int max_total, const float* params, hipStream_t stream);
// max |w| of every job with a wmax word (run in front of pack_jobs_launch, on words the caller has zeroed: the maximum of
// THIS pack).  status != nullptr: bit 0 (kStatusFwdStats) is OR-ed in when a maximum is not finite
int wabsmax_jobs_launch(const PackJob* jobs_dev, int njobs, int max_total, const float* params, hipStream_t stream,
                        int* status = nullptr);

// weight packing: torch OIHW -> [9][rows_pad][cols] (see conv3x3.hip)
int pack_weights_launch(const float* w, float* dst, int cout, int cin, int rows_pad, int cols,
                        const int* row_map, const int* col_map, int transposed, hipStream_t stream);

}  // namespace mimo
