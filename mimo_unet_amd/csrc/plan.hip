// mimo_plan: the MIMO U-Net network executor behind the C ABI.  Owns packed weights, saved
// activations, BatchNorm statistics and scratch; sequences the HIP kernels of conv3x3.hip /
// elementwise.hip for forward, loss and backward on the caller's stream.
//
// Topology restated from the reference (relative to /root/reference):
//   MimoUNet.forward ........... mimo/models/mimo_components/model.py:94-117
//   SubnetworkEncoder .......... model.py:119-175   (S private DoubleConv + Down)
//   SubnetworkCore ............. model.py:178-243   (down2..4, up1..3 on the channel concat)
//   SubnetworkDecoder .......... model.py:246-297   (S private Up + OutConv, stacked)
//   DoubleConv / Down / Up ..... mimo/models/mimo_components/components.py:8-120
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "elementwise.h"

namespace mimo {

static thread_local char g_error[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_error; }

namespace {

struct TensorInfo {
  std::string name;
  int ndim;
  int64_t shape[4];
  int kind;  // 0 parameter, 1 BN buffer
  int64_t offset;
};

// An activation tensor as seen by consumers: pointer (possibly a channel slice of a wider
// concat buffer), pixel pitch, padded channel count, gradient buffer, and the map from padded
// channel index to the logical channel index inside this tensor (-1 = zero padding).
struct Act {
  float* a = nullptr;
  int ld = 0;
  float* da = nullptr;
  int ldda = 0;
  int Cp = 0, C = 0;
  int N = 0, H = 0, W = 0;
  std::vector<int> chmap;
  int grad_writes = 0;  // run-time counter: first writer assigns, later writers accumulate
  // skip-connection gradient of this tensor, left in the consuming Up block's own padded-domain buffer (channels
  // [0, Cp), pixel pitch skipgrad_ld) until the MaxPool backward of the same tensor folds it in
  const float* skipgrad = nullptr;
  int skipgrad_ld = 0;
  bool pooled = false;  // some Down block pools this tensor (its pool_bwd then takes the skip gradient along)
  // Pooled gradient of this tensor left in the pooling Down block's own padded-domain buffer (pixel pitch poolgrad_ld) for
  // the BatchNorm backward of the producing convolution(s) to route through the max-pool windows themselves (GS_POOL,
  // elementwise.h): pool_bwd is not launched and `da` is not written.  A channel slice of a concat tensor (down1[s] inside
  // x2cat) finds both gradients on `parent`, at channel offset parent_choff.
  const float* poolgrad = nullptr;
  int poolgrad_ld = 0;
  Act* parent = nullptr;
  int parent_choff = 0;
  // != nullptr: in a TRAINING forward `a` is not written — its readers (the bilinear up-sampling into the next block's
  // concat buffer, the 1x1 head and its backward) take the producing convolution's pre-activation tensor z (pixel pitch
  // z_ld) and apply scale / shift + ReLU themselves (round 4; Up blocks without dropout in split16)
  const float *z = nullptr, *z_scale = nullptr, *z_shift = nullptr;
  int z_ld = 0;
  // per forward call: the readers really go through z (set by dc_forward).  False — and `a` is written by the
  // BatchNorm + ReLU pass as on any other block — whenever a multiplier acts on the activated tensor in this call: a
  // caller-supplied Dropout2d mask on a block built with rate 0 (mimo_forward_args.drop_masks is honoured on every
  // block), or an element-wise final-dropout mask on a decoder output
  bool z_live = false;
};

struct ConvBN {
  std::string conv_name, bn_name;
  int Cin = 0, Cout = 0, cin_p = 0, cout_p = 0;
  int cout_pad = 0;             // forward packed rows
  int dg_rows = 0;              // dgrad packed rows (covers cin_p)
  int wg_cin_pad = 0, wg_cout_pad = 0, wg_splits = 1;
  int N = 0, H = 0, W = 0;
  int64_t off_w = 0, off_b = 0, off_gamma = 0, off_beta = 0, off_rm = 0, off_rv = 0;
  float *wf = nullptr, *wd = nullptr, *bias_p = nullptr;
  void *wf16 = nullptr, *wd16 = nullptr;  // split-bf16 packed weights (MIMO_PREC_SPLIT16)
  bool fwd_split = false, dg_split = false, wg_split = false;
  unsigned* wmax = nullptr;  // fp16 forward weight image: max |w| of the layer AT ITS LAST PACK, float bits (w16_scale: the image's power-of-two scale)
  bool wg_np2 = false;  // the weight gradient runs two fp16 MFMAs per product (wgrad_split.hip NP == 2)
  bool thin = false;  // image convolution (<= 4 input channels) on the plain-FMA kernels of conv_thin.hip
  int fwd_wide = 0, dg_wide = 0;  // != 0: the launch runs on conv_wide.hip, value = packed weight rows (conv3x3_wide_rows)
  int *cin_map = nullptr, *fwd_row_map = nullptr, *dg_row_map = nullptr, *dg_col_map = nullptr;
  float* z = nullptr;
  int dtz = ST_F32;  // element type of z: the plan's storage type, fp32 where the fp32 kernel family writes it
  float *mean = nullptr, *invstd = nullptr, *scale = nullptr, *shift = nullptr, *c1 = nullptr, *c2 = nullptr;
  const float* in = nullptr;
  int ld_in = 0;
  float* a = nullptr;  // output activation
  int ld_a = 0;
  float* pool_out = nullptr;  // the MaxPool2d(2) of `a` is written by the BatchNorm + ReLU pass (training forward)
  int pool_ld = 0;
  // Second convolution of a DoubleConv whose loaders apply the first one's BatchNorm + ReLU themselves (round 4): forward and
  // weight gradient read the first convolution's pre-activation tensor `in_z` with its scale / shift, and the activated
  // tensor between the two convolutions (components.py:24-25) is never written.  `act_elided` marks that first convolution.
  bool fuse_in = false, act_elided = false;  // (act_elided on a second convolution: its readers apply it, see Act::z)
  const float* in_z = nullptr;
  int ld_in_z = 0;
  const float *in_scale = nullptr, *in_shift = nullptr;
};

enum InputKind { IN_IMAGE = 0, IN_POOL = 1, IN_UPCAT = 2 };

struct DoubleConv {
  std::string prefix;
  ConvBN c1, c2;
  InputKind kind = IN_IMAGE;
  int subnet = -1;       // for IN_IMAGE
  Act* src0 = nullptr;   // IN_POOL: pooled tensor; IN_UPCAT: skip
  Act* src1 = nullptr;   // IN_UPCAT: low-resolution tensor
  float* in_buf = nullptr;  // materialised input (packed image / pooled / concat)
  int in_ld = 0;
  bool skip_in_place = false;  // IN_UPCAT: the skip tensor already lives in channels [0, Cs) of in_buf
  bool pool_fused = false;     // IN_POOL: in_buf is written by the producers' BatchNorm + ReLU pass
  float* dxpad_own = nullptr;  // IN_UPCAT: private buffer of c1's padded-domain data gradient (holds the skip slice
                               // until the skip tensor's pool backward has read it)
  float* mid = nullptr;  // a1
  Act out;               // a2 (+ gradient)
  float drop_p = 0.f;
  const float* mask = nullptr;  // set per forward call
  int64_t p_begin = 0, p_end = 0;  // this block's parameters in the flat parameter / gradient buffers (floats)
};

struct Head {
  int s = 0;
  int64_t off_w = 0, off_b = 0;
};

}  // namespace
}  // namespace mimo

using namespace mimo;

struct mimo_plan {
  mimo_config cfg;
  int S, f, N, H, W, Ci, Co, Ci_p;
  // 16-bit storage modes (MIMO_PREC_BF16_MIXED / MIMO_PREC_FP16_MIXED): activations, conv outputs and their gradients
  // live in HBM as bf16 / fp16 (st, esz bytes per element); every "float*" of such a tensor is then an untyped byte
  // address and channel offsets go through eoff()
  bool mixed = false, f16 = false;
  int st = ST_F32, esz = 4;
  float* eoff(float* p, size_t elems) const { return reinterpret_cast<float*>(reinterpret_cast<char*>(p) + elems * esz); }
  int alloc_act(float** p, size_t elems, int dt) {  // activation-like tensor of `elems` elements of StoreType dt
    return dalloc(p, (elems * store_bytes(dt) + 3) / 4);
  }
  int fwd_mode() const {  // conv3x3_bf16x3_launch modes
    return cfg.precision == MIMO_PREC_BF16 ? 2 : cfg.precision == MIMO_PREC_BF16_MIXED ? 4 : cfg.precision == MIMO_PREC_FP16_MIXED ? 6 : 1;
  }
  int dgrad_mode() const {
    return cfg.precision == MIMO_PREC_BF16 ? 3 : cfg.precision == MIMO_PREC_BF16_MIXED ? 5 : cfg.precision == MIMO_PREC_FP16_MIXED ? 7 : 0;
  }
  std::vector<TensorInfo> tensors;
  int64_t param_floats = 0, buffer_floats = 0;
  float *params = nullptr, *grads = nullptr, *bnbuf = nullptr;
  std::vector<void*> allocs;
  size_t bytes = 0;

  std::vector<std::unique_ptr<DoubleConv>> dcs;  // forward order == oracle double_conv_specs
  std::vector<DoubleConv*> enc_in, down1, up4;
  DoubleConv *down2 = nullptr, *down3 = nullptr, *down4 = nullptr, *up1 = nullptr, *up2 = nullptr, *up3 = nullptr;
  Act x2cat;  // concat of the S encoder outputs (model.py:113)
  std::vector<Head> heads;

  // scratch
  float *s_dz = nullptr, *s_dxpadA = nullptr, *s_dxpadB = nullptr, *s_wslab = nullptr,
        *s_partial = nullptr, *s_losspart = nullptr;
  float* s_headpart = nullptr;  // GS_HEAD: the head's weight / bias gradient partial rows, written by the BatchNorm-backward reduction
  int head_rows = 0;            // ... and how many (that launch's grid)
  double* s_sums = nullptr;
  int* s_tickets = nullptr;  // colsum tickets of the scratch set in use (zero between launches)
  ColsumScratch colsum() const { return ColsumScratch{s_sums, s_tickets, d_status}; }
  int* d_status = nullptr;  // numerics status word (mimo_plan_status)
  size_t cap_act = 0, cap_pad = 0, cap_slab = 0, cap_partial = 0, cap_sums = 0;
  float* s_kpart = nullptr;  // partial-sum slabs of the K-split convolution launches (conv3x3_bf16x3_launch_k)
  size_t cap_kpart = 0;

  // MIMO_WGRAD_STREAM (default 1): weight gradients on a side stream — wgrad(L) (matrix-pipe bound, little HBM
  // traffic) overlaps the bandwidth-bound BatchNorm / gather kernels of the layers below it on the caller's stream;
  // dz ping-pongs between two buffers so that layer L-1 can write its dz while wgrad(L) still reads the other one.
  // (Measured and removed in round 3: per-layer dz buffers without back-pressure, 3-4 ping-pong buffers, releasing a
  // weight gradient only after its layer's data gradient — none faster)
  bool wg_async = false;
  int wg_cus = 256;  // CUs the weight-gradient launches are sized for (sched::wg_side_cus; MIMO_WGRAD_CUS overrides)
  // test hook (MIMO_DEBUG_WGRAD_DELAY_US, read per plan): an idle kernel of that many microseconds in front of every weight
  // gradient, on the stream it runs on — the consumer of dz and its max |dz| slots arrives late (tests/test_streams_gpu.py)
  int wg_delay_us = 0;
  // the hand-off events between the two streams ride on the launches that produce what they announce (hipExtLaunchKernelGGL
  // stop event) instead of hipEventRecord calls behind them: MIMO_EVENT_ON_LAUNCH=0 restores the records (A/B; read per plan)
  bool ev_attach = true;
  // BatchNorm backward forms the gradient arriving at a pooled tensor / at the head's input itself (GS_POOL / GS_HEAD):
  // fp32 storage, MIMO_FUSE_BWD_SRC=0 switches it off (read per plan)
  bool fuse_bwd_src = false;
  bool fuse_bwd_pool = false, fuse_bwd_head = false;  // (MIMO_FUSE_BWD_SRC=2: pooled tensors only, 3: head only — A/B)
  hipStream_t wg_stream = nullptr;
#ifndef MIMO_DZ_BUFS
// dz buffers (with their max |dz| slots) the side stream's weight gradients may lag behind.  Round 6: 4 buffers, released in
// PAIRS — the main stream waits for the side stream once per two layers (in front of an even buffer, for the event of the odd
// one behind it: the side stream is in order, so that covers both) instead of once per layer.  A cross-stream wait in front
// of a kernel costs the waiting stream ~4 us on this stack even when it is already satisfied (scripts/micro/event_cost.hip);
// 2 buffers with a wait per layer (rounds 2-5) remain as the A/B build -DMIMO_DZ_BUFS=2.
#define MIMO_DZ_BUFS 4
#endif
  static constexpr int kDzBufs = MIMO_DZ_BUFS;
  static constexpr int wg_bufs = kDzBufs;
  hipEvent_t ev_dz[kDzBufs] = {}, ev_wg[kDzBufs] = {}, ev_join = nullptr, ev_stage = nullptr;
  bool wg_pending[kDzBufs] = {};
  float* s_dz2[kDzBufs] = {};
  float* s_dzmax2[kDzBufs] = {};  // per-workgroup maxima of |dz| of the tensor in s_dz2[i] (two-MFMA weight gradient)
  // split (bf16 hi|lo) copy of dz for layers whose data gradient runs on the fp32 kernel while the weight
  // gradient runs on the bf16-pair kernel (fewer than 16 output channels); null when no layer needs it
  float* s_dzs2[kDzBufs] = {};
  bool any_mixed_dz = false;
  int dz_idx = 0;

  // optional per-kernel-class timing with HIP events on the launch stream (bench.py roofline)
  struct ProfRec {
    hipEvent_t a, b;
    int kind, tier;
  };
  int cur_tier = 0;  // resolution tier of the block being issued (kernel-class records inherit it)
  double prof_kind_tier_ms[MIMO_PROF_KINDS][5] = {};
  bool prof_on = false;
  std::vector<ProfRec> prof_recs;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;
  double prof_ms[MIMO_PROF_KINDS] = {}, prof_flops[MIMO_PROF_KINDS] = {}, prof_bytes[MIMO_PROF_KINDS] = {};
  int64_t prof_launches[MIMO_PROF_KINDS] = {};

  // tier records: kind = kProfTierBase + 2 * tier + (backward ? 1 : 0); tier = resolution level of a DoubleConv
  // (0 = full resolution ... 4 = 1/16): the whole block (every launch between the two events), so that a tier's
  // summed device time can be priced against its algorithmic HBM bytes
  static constexpr int kProfTierBase = 100, kProfTiers = 5;
  double prof_tier_ms[2 * kProfTiers] = {};
  int tier_of(int h) const {
    int t = 0, hh = H;
    while (t < kProfTiers - 1 && hh > h) {
      hh /= 2;
      ++t;
    }
    return t;
  }
  int prof_begin(int kind, hipStream_t st) {
    if (!prof_on) return -1;
    ProfRec r;
    if (!prof_pool.empty()) {
      r.a = prof_pool.back().first;
      r.b = prof_pool.back().second;
      prof_pool.pop_back();
    } else {
      (void)hipEventCreate(&r.a);
      (void)hipEventCreate(&r.b);
    }
    r.kind = kind;
    if (kind >= kProfTierBase) cur_tier = (kind - kProfTierBase) / 2;
    r.tier = cur_tier;
    (void)hipEventRecord(r.a, st);
    prof_recs.push_back(r);
    return (int)prof_recs.size() - 1;
  }
  void prof_end(int idx, double flops, double bytes, hipStream_t st) {
    if (idx < 0 || !prof_on) return;
    (void)hipEventRecord(prof_recs[idx].b, st);
    const int kind = prof_recs[idx].kind;
    if (kind >= kProfTierBase) return;
    prof_flops[kind] += flops;
    prof_bytes[kind] += bytes;
    prof_launches[kind] += 1;
  }
  int prof_collect() {
    for (auto& r : prof_recs) {
      MIMO_HIP_CHECK(hipEventSynchronize(r.b));
      float ms = 0.f;
      MIMO_HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
      if (r.kind >= kProfTierBase)
        prof_tier_ms[r.kind - kProfTierBase] += ms;
      else {
        prof_ms[r.kind] += ms;
        prof_kind_tier_ms[r.kind][r.tier] += ms;
      }
      prof_pool.emplace_back(r.a, r.b);
    }
    prof_recs.clear();
    return MIMO_OK;
  }

  // hipGraph replay of the eval-mode forward (launch-bound at small batch: ~100+ kernel launches).
  // The captured kernels read plan-owned staging copies of x / perm / masks and write a plan-owned
  // logits buffer, so the graph stays valid when the caller's tensors move.
  bool graph_enabled = true;
  hipStream_t cap_stream = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  uint64_t graph_key = 0;
  // hipGraph replay of the TRAINING step (round 6, MIMO_TRAIN_GRAPH=1, opt-in; mimo_unet.py:115-144 at its per-GPU shard is ~290
  // launches of which ~200 run < 20 us): the training forward is one graph, the backward one graph (mimo_backward) or one
  // per stage (mimo_backward_stage: the data-parallel caller starts a stage's all-reduce between two graphs).  Captured
  // kernels only see plan-owned memory: x / perm / Dropout2d multipliers are staged in front of the forward graph, label /
  // mask / perm by mimo_loss_forward, dloss in front of the first backward graph, the logits leave through g_out.  A call
  // shape (key) is captured the SECOND time it is seen — one-off shapes stay eager, and every kernel has run eagerly once
  // before it is captured.  The side stream's weight gradients are a fork / join inside each backward graph.
  bool train_graph = false;
  hipGraphExec_t tg_fwd = nullptr;
  uint64_t tg_fwd_key = 0, tg_fwd_seen = 0;
  static constexpr int kBwdGraphs = 9;  // [0, 8): the single stages; [8]: the whole backward
  hipGraphExec_t tg_bwd[kBwdGraphs] = {};
  uint64_t tg_bwd_key = 0, tg_bwd_seen = 0;
  bool tg_bwd_live = false;    // the backward in progress replays graphs (decided at its stage 0)
  // captures this plan may still make: a caller that keeps alternating call shapes on one plan would otherwise re-capture
  // (milliseconds of host time) every other step
  static constexpr int kMaxTrainCaptures = 24;
  int tg_captures = 0;
  bool fwd_graphed = false;    // the last forward was a training-graph replay: logits in g_out, masks staged
  bool loss_staged = false;    // ... and mimo_loss_forward staged label / mask / perm
  int64_t last_x_rows = 0;     // rows of the last forward's x (= rows of the label / mask tensors of that batch)
  const int64_t* last_perm_arg = nullptr;  // the caller's perm tensor of the last graphed forward (staged in g_perm)
  float *g_label = nullptr, *g_lmask = nullptr, *g_dloss = nullptr;
  int64_t* g_lperm = nullptr;
  void drop_train_graphs() {
    if (tg_fwd) (void)hipGraphExecDestroy(tg_fwd);
    tg_fwd = nullptr;
    tg_fwd_key = tg_fwd_seen = 0;
    for (auto& e : tg_bwd) {
      if (e) (void)hipGraphExecDestroy(e);
      e = nullptr;
    }
    tg_bwd_key = tg_bwd_seen = 0;
  }
  void drop_graphs() {
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    graph_exec = nullptr;
    drop_train_graphs();
  }
  float *g_x = nullptr, *g_out = nullptr;
  int64_t* g_perm = nullptr;
  std::vector<float*> g_masks;
  std::vector<const float*> g_mask_ptrs;
  // in-engine dropout (mimo_forward_args.rng_sites): Dropout2d multipliers are drawn into g_masks by one Philox
  // launch per forward; the element-wise sites (center / final nn.Dropout) regenerate theirs on the fly in the
  // forward and in the backward from the (seed, offset) of the last forward
  Dropout2dSite* d_sites = nullptr;
  int max_site_count = 0;
  uint64_t rng_seed = 0, rng_offset = 0;
  std::vector<char> elem_rng_on;  // [1 + S]
  std::vector<const float*> eff_masks;
  float elem_rate(int j) const { return j == 0 ? cfg.center_dropout_rate : cfg.final_dropout_rate; }
  ElemRng elem_rng(int j) const { return ElemRng{rng_seed, rng_offset, (int)dcs.size() + j, elem_rate(j)}; }

  bool capturing = false;

  // max |w| words of the fp16 forward weight images (ConvBN::wmax), one array: zeroed and re-taken by every pack, so that a
  // transient huge weight (a diverged step followed by load_state_dict) does not leave a layer's image at a tiny scale
  static constexpr int kMaxWmaxWords = 128;
  unsigned* wmax_pool = nullptr;
  int wmax_used = 0;
  // weight repack job tables (device): [0, n_fwd_jobs) forward packs (+ bias copies), then the data-gradient packs
  PackJob* pack_jobs = nullptr;
  int n_fwd_jobs = 0, n_all_jobs = 0, pack_max_total = 0;

  // per-call state
  bool fwd_done = false, fwd_training = false, had_perm = false, loss_done = false;
  int bwd_next_stage = 0;  // staged backward: the stage that may run next (0 = a fresh backward)
  bool fwd_no_grad = false;       // last forward folded BN/ReLU into the conv epilogue: nothing saved for a backward
  int64_t derived_version = -1;   // param_version the packed weights / eval scale+shift were derived from (-1: none)
  bool derived_dgrad = false;     // ... including the data-gradient weight copies
  bool need_derive = true;        // per-call: (re)pack weights and eval BN constants in this forward
  int64_t encoder_param_floats = 0;
  float* out = nullptr;
  const float *label = nullptr, *lmask = nullptr;
  std::vector<const float*> elem_masks;  // [1 + S] element-wise dropout multipliers of the last forward (or empty)
  const int64_t* lperm = nullptr;

  ~mimo_plan() {
    for (void* p : allocs) (void)hipFree(p);
    for (auto& r : prof_recs) {
      (void)hipEventDestroy(r.a);
      (void)hipEventDestroy(r.b);
    }
    for (auto& e : prof_pool) {
      (void)hipEventDestroy(e.first);
      (void)hipEventDestroy(e.second);
    }
    drop_graphs();
    if (cap_stream) (void)hipStreamDestroy(cap_stream);
    if (wg_stream) (void)hipStreamDestroy(wg_stream);
    for (int i = 0; i < kDzBufs; ++i)
      for (hipEvent_t e : {ev_dz[i], ev_wg[i]})
        if (e) (void)hipEventDestroy(e);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (ev_stage) (void)hipEventDestroy(ev_stage);
  }

  template <typename T>
  int dalloc(T** p, size_t count) {
    void* q = nullptr;
    const size_t nbytes = std::max<size_t>(count, 1) * sizeof(T);
    MIMO_HIP_CHECK(hipMalloc(&q, nbytes));
    MIMO_HIP_CHECK(hipMemset(q, 0, nbytes));
    allocs.push_back(q);
    bytes += nbytes;
    *p = static_cast<T*>(q);
    return MIMO_OK;
  }
  int upload_ints(int** p, const std::vector<int>& v) {
    MIMO_TRY(dalloc(p, v.size()));
    MIMO_HIP_CHECK(hipMemcpy(*p, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
    return MIMO_OK;
  }

  int64_t add_tensor(const std::string& name, std::vector<int64_t> shape, int kind) {
    TensorInfo t;
    t.name = name;
    t.ndim = (int)shape.size();
    int64_t n = 1;
    for (int i = 0; i < 4; ++i) {
      t.shape[i] = i < t.ndim ? shape[i] : 1;
      n *= t.shape[i];
    }
    t.kind = kind;
    int64_t& cur = kind == 0 ? param_floats : buffer_floats;
    t.offset = cur;
    cur += (n + 3) / 4 * 4;  // keep every tensor 16-byte aligned inside the flat buffers
    tensors.push_back(t);
    return t.offset;
  }

  int init_convbn(ConvBN& L, const std::string& prefix, int conv_idx, int bn_idx, int Cin, int Cout,
                  const std::vector<int>& in_chmap, int n, int h, int w) {
    L.conv_name = prefix + "." + std::to_string(conv_idx);
    L.bn_name = prefix + "." + std::to_string(bn_idx);
    L.Cin = Cin;
    L.Cout = Cout;
    L.cin_p = (int)in_chmap.size();
    L.cout_p = pad_channels(Cout);
    L.cout_pad = conv3x3_cout_pad(Cout);
    L.dg_rows = conv3x3_cout_pad(L.cin_p);
    L.N = n;
    L.H = h;
    L.W = w;
    const bool mfma16 = cfg.precision != MIMO_PREC_FP32;
    L.wg_split = mfma16;
    if (L.wg_split) {
      int CI, CO;
      wgrad_split_tiles(L.cin_p, L.cout_p, &CI, &CO);
      L.wg_cin_pad = round_up(L.cin_p, CI);
      L.wg_cout_pad = round_up(L.cout_p, CO);
      // operand storage of this layer's weight gradient (= WgradLaunch::store in convbn_backward: fwd_split below)
      const int wg_store = !mixed ? 0 : (L.cin_p >= 8 ? (f16 ? 2 : 1) : (f16 ? 4 : 3));
      L.wg_splits = wgrad_split_pick_splits(n, h, w, L.wg_cin_pad, L.wg_cout_pad, CI, CO, wg_store, wg_cus);
    } else {
      L.wg_cin_pad = round_up(L.cin_p, 32);
      L.wg_cout_pad = round_up(L.cout_p, 32);
      L.wg_splits = wgrad_pick_splits(n, h, w, L.wg_cin_pad, L.wg_cout_pad);
    }
    L.off_w = add_tensor(L.conv_name + ".weight", {Cout, Cin, 3, 3}, 0);
    L.off_b = add_tensor(L.conv_name + ".bias", {Cout}, 0);
    L.off_gamma = add_tensor(L.bn_name + ".weight", {Cout}, 0);
    L.off_beta = add_tensor(L.bn_name + ".bias", {Cout}, 0);
    L.off_rm = add_tensor(L.bn_name + ".running_mean", {Cout}, 1);
    L.off_rv = add_tensor(L.bn_name + ".running_var", {Cout}, 1);
    const bool train_bufs = !cfg.inference_only;  // everything only a backward reads or writes
    MIMO_TRY(dalloc(&L.wf, (size_t)9 * L.cout_pad * L.cin_p));
    if (train_bufs) MIMO_TRY(dalloc(&L.wd, (size_t)9 * L.dg_rows * L.cout_p));
    MIMO_TRY(dalloc(&L.bias_p, L.cout_pad));
    // split-bf16 MFMA needs a K chunk of 32 channels; the 2..4-channel image conv stays on the fp32 kernel
    // (16-bit storage: every layer but the image convolution runs on the 16-bit kernels, whose loaders move 8-channel units)
    L.fwd_split = mfma16 && L.cin_p >= (mixed ? 8 : 16);
    L.dg_split = mfma16 && (mixed || L.cout_p >= 16);
    L.dtz = L.fwd_split ? st : ST_F32;
    {
      bool ident = true;  // (the packed image: logical channel i in padded channel i)
      for (int i = 0; i < Cin; ++i) ident = ident && i < (int)in_chmap.size() && in_chmap[i] == i;
      L.thin = !L.fwd_split && ident && conv3x3_thin_ok(Cin, L.cout_p);
      if (L.thin && train_bufs) cap_slab = std::max(cap_slab, wgrad_thin_scratch(Cin, L.cout_p));
    }
    if (L.wg_split && !L.dg_split) any_mixed_dz = true;
    // two fp16 MFMAs per product: not for the image convolution (raw inputs need not fit fp16: its forward runs on the fp32
    // kernels), and only while the launch that writes dz has a max |dz| slot for each of its workgroups
    // (<= 2048 x ceil(Cp / 1024); wider layers keep the three-MFMA arithmetic)
    L.wg_np2 = cfg.precision == MIMO_PREC_SPLIT16 && L.wg_split && train_bufs && L.fwd_split && wgrad_split_has_np2(L.cin_p, L.cout_p) &&
               2048 * ceil_div(L.cout_p / 4, 256) <= kDzMaxSlots;
    if (cfg.precision == MIMO_PREC_SPLIT16 || mixed) {  // decomposition per layer and direction (sched::wide_config)
      if (L.fwd_split) L.fwd_wide = conv3x3_wide_rows(fwd_mode(), n, L.cin_p, L.cout_p, h, w);
      if (L.dg_split) L.dg_wide = conv3x3_wide_rows(dgrad_mode(), n, L.cout_p, L.cin_p, h + 2, w + 2);
    }
    if (cfg.precision == MIMO_PREC_SPLIT16) {  // K split of the few-tile / long-K launches (conv3x3_ksplit): scratch for the slabs
      if (L.fwd_split && !L.fwd_wide && !cfg.inference_only)
        cap_kpart = std::max(cap_kpart, conv3x3_ksplit_scratch(fwd_mode(), n, L.cin_p, L.cout_pad, h, w, L.cout_p));
      if (L.dg_split && !L.dg_wide && train_bufs)
        cap_kpart = std::max(cap_kpart, conv3x3_ksplit_scratch(dgrad_mode(), n, L.cout_p, L.dg_rows, h + 2, w + 2, L.cin_p));
    }
    if (L.fwd_split && (cfg.precision == MIMO_PREC_SPLIT16 || cfg.precision == MIMO_PREC_FP16_MIXED)) {
      // one array for all layers' words: every pack zeroes it in one memset and takes the maxima afresh (pack_all)
      if (!wmax_pool) MIMO_TRY(dalloc(&wmax_pool, kMaxWmaxWords));
      if (wmax_used >= kMaxWmaxWords) {
        set_error("too many convolution layers (%d) for the fp16 weight-scale words", wmax_used + 1);
        return MIMO_ERR_INVALID;
      }
      L.wmax = wmax_pool + wmax_used++;
    }
    if (L.fwd_split) {
      uint16_t* q = nullptr;
      MIMO_TRY(dalloc(&q, L.fwd_wide ? conv3x3_wide_weight_elems(L.cin_p, L.fwd_wide)
                                     : (size_t)ceil_div(L.cin_p, 32) * 9 * L.cout_pad * 64));
      L.wf16 = q;
    }
    if (L.dg_split && train_bufs) {
      uint16_t* q = nullptr;
      MIMO_TRY(dalloc(&q, L.dg_wide ? conv3x3_wide_weight_elems(L.cout_p, L.dg_wide)
                                    : (size_t)ceil_div(L.cout_p, 32) * 9 * L.dg_rows * 64));
      L.wd16 = q;
    }
    MIMO_TRY(upload_ints(&L.cin_map, in_chmap));
    std::vector<int> frm(L.cout_pad), drm(L.dg_rows), dcm(L.cout_p);
    for (int i = 0; i < L.cout_pad; ++i) frm[i] = i < Cout ? i : -1;
    for (int i = 0; i < L.dg_rows; ++i) drm[i] = i < L.cin_p ? in_chmap[i] : -1;
    for (int i = 0; i < L.cout_p; ++i) dcm[i] = i < Cout ? i : -1;
    MIMO_TRY(upload_ints(&L.fwd_row_map, frm));
    MIMO_TRY(upload_ints(&L.dg_row_map, drm));
    MIMO_TRY(upload_ints(&L.dg_col_map, dcm));
    // (16-bit storage: the image convolution runs on the fp32 kernels and always goes through an fp32 z)
    if (train_bufs || (mixed && !L.fwd_split)) MIMO_TRY(alloc_act(&L.z, (size_t)n * h * w * L.cout_p, L.dtz));
    MIMO_TRY(dalloc(&L.mean, L.cout_p));
    MIMO_TRY(dalloc(&L.invstd, L.cout_p));
    MIMO_TRY(dalloc(&L.scale, L.cout_p + kFinSlack));  // (+ zero slack: the next convolution's loaders read whole K chunks)
    MIMO_TRY(dalloc(&L.shift, L.cout_p + kFinSlack));
    MIMO_TRY(dalloc(&L.c1, L.cout_p));
    MIMO_TRY(dalloc(&L.c2, L.cout_p));
    const size_t act = (size_t)n * h * w * L.cout_p;
    const size_t padv = (size_t)n * (h + 2) * (w + 2) * L.cin_p;
    cap_act = std::max(cap_act, act);
    cap_pad = std::max(cap_pad, padv);
    // slabs + the reduction's group-sum levels (geometric: < splits/15 extra slabs)
    cap_slab = std::max(cap_slab, (size_t)(L.wg_splits + L.wg_splits / 8 + 2) * 9 * L.wg_cin_pad * L.wg_cout_pad);
    const size_t stat_rows = std::max(conv3x3_stat_rows(n, h, w), conv3x3_ws_stat_rows(n, h, w));
    cap_partial = std::max(cap_partial, stat_rows * 2 * L.cout_pad);
    cap_partial = std::max(cap_partial, (size_t)kBnReduceMaxBlocks * 2 * L.cout_p);
    cap_sums = std::max(cap_sums, (size_t)kMaxChunks * 2 * round_up(std::max(L.cout_pad, L.cout_p), 64));
    return MIMO_OK;
  }

  // Create a DoubleConv whose input has channel map in_chmap at resolution (h, w).
  int make_dc(DoubleConv** outp, const std::string& prefix, const std::vector<int>& in_chmap, int Cin, int Cmid,
              int Cout, int h, int w, float drop_p, float* out_a, int out_ld, float* out_da, int out_ldda) {
    auto dc = std::make_unique<DoubleConv>();
    dc->prefix = prefix;
    dc->drop_p = drop_p;
    dc->p_begin = param_floats;
    MIMO_TRY(init_convbn(dc->c1, prefix, 0, 1, Cin, Cmid, in_chmap, N, h, w));
    std::vector<int> midmap(pad_channels(Cmid));
    for (size_t i = 0; i < midmap.size(); ++i) midmap[i] = (int)i < Cmid ? (int)i : -1;
    MIMO_TRY(init_convbn(dc->c2, prefix, 3, 4, Cmid, Cout, midmap, N, h, w));
    dc->p_end = param_floats;
    MIMO_TRY(alloc_act(&dc->mid, (size_t)N * h * w * dc->c1.cout_p, st));
    Act& o = dc->out;
    o.N = N;
    o.H = h;
    o.W = w;
    o.C = Cout;
    o.Cp = dc->c2.cout_p;
    o.chmap.resize(o.Cp);
    for (int i = 0; i < o.Cp; ++i) o.chmap[i] = i < Cout ? i : -1;
    if (out_a) {
      o.a = out_a;
      o.ld = out_ld;
      o.da = out_da;
      o.ldda = out_ldda;
    } else {
      MIMO_TRY(alloc_act(&o.a, (size_t)N * h * w * o.Cp, st));
      if (!cfg.inference_only) MIMO_TRY(alloc_act(&o.da, (size_t)N * h * w * o.Cp, st));
      o.ld = o.ldda = o.Cp;
    }
    dc->c1.a = dc->mid;
    dc->c1.ld_a = dc->c1.cout_p;
    dc->c2.in = dc->mid;
    dc->c2.ld_in = dc->c1.cout_p;
    {
      // BatchNorm + ReLU of the first convolution applied by the second one's loaders: split16, when BOTH readers of the
      // activated tensor — the second convolution's forward and its weight gradient — run on kernels that can (the
      // wide / 256-pixel wave-specialised forward, the wave-specialised weight gradient).  MIMO_FUSE_BN_IN=0: materialise.
      const bool on = !(getenv("MIMO_FUSE_BN_IN") && atoi(getenv("MIMO_FUSE_BN_IN")) == 0);  // read per plan (A/B, tests)
      ConvBN& c2 = dc->c2;
      if (on && cfg.precision == MIMO_PREC_SPLIT16 && !cfg.inference_only && c2.fwd_split && c2.wg_split && dc->c1.dtz == ST_F32 &&
          conv3x3_split_fuses_input(fwd_mode(), c2.fwd_wide, h, w) && wgrad_split_fuses_input(c2.cin_p, c2.cout_p, 0, 3)) {
        c2.fuse_in = true;
        c2.in_z = dc->c1.z;
        c2.ld_in_z = dc->c1.cout_p;
        c2.in_scale = dc->c1.scale;
        c2.in_shift = dc->c1.shift;
        dc->c1.act_elided = true;
      }
    }
    dc->c2.a = o.a;
    dc->c2.ld_a = o.ld;
    *outp = dc.get();
    dcs.push_back(std::move(dc));
    return MIMO_OK;
  }

  // Skip connections cost no copy: the skip tensor is re-homed into channels [0, Cs) of the concat buffer of
  // the Up block that consumes it (its producers write it there with the concat's pixel pitch), so the
  // up-sample + concat kernel only writes the up-sampled channels.  `producers`: the ConvBN layers whose `a`
  // is (a channel slice of) this tensor, with their channel offsets.
  void rehome_skip(Act& sk, DoubleConv* up, std::vector<std::pair<DoubleConv*, int>> producers) {
    sk.a = up->in_buf;
    sk.ld = up->in_ld;
    for (auto& pr : producers) {
      DoubleConv* dc = pr.first;
      dc->c2.a = eoff(up->in_buf, pr.second);
      dc->c2.ld_a = up->in_ld;
      dc->out.a = dc->c2.a;
      dc->out.ld = up->in_ld;
    }
    up->skip_in_place = true;
  }

  int set_input(DoubleConv* dc, InputKind kind, Act* s0, Act* s1, int in_cp, int h, int w) {
    dc->kind = kind;
    dc->src0 = s0;
    dc->src1 = s1;
    MIMO_TRY(alloc_act(&dc->in_buf, (size_t)N * h * w * in_cp, kind == IN_IMAGE ? ST_F32 : st));  // the packed image stays fp32
    if (kind == IN_POOL) s0->pooled = true;
    static const bool skip_keep = !(getenv("MIMO_SKIP_GRAD_IN_PLACE") && atoi(getenv("MIMO_SKIP_GRAD_IN_PLACE")) == 0);
    if (kind == IN_UPCAT && skip_keep && s0->pooled && !cfg.inference_only)
      MIMO_TRY(alloc_act(&dc->dxpad_own, (size_t)N * (h + 2) * (w + 2) * in_cp, st));
    // the pooled gradient stays in this block's own buffer until the producers' BatchNorm backward has routed it (GS_POOL)
    if (kind == IN_POOL && fuse_bwd_pool) MIMO_TRY(alloc_act(&dc->dxpad_own, (size_t)N * (h + 2) * (w + 2) * in_cp, st));
    dc->in_ld = in_cp;
    dc->c1.in = dc->in_buf;
    dc->c1.ld_in = in_cp;
    return MIMO_OK;
  }

  static std::vector<int> cat_map(const Act& a, const Act& b) {
    std::vector<int> m = a.chmap;
    for (int v : b.chmap) m.push_back(v < 0 ? -1 : v + a.C);
    return m;
  }

  int build() {
    mixed = cfg.precision == MIMO_PREC_BF16_MIXED || cfg.precision == MIMO_PREC_FP16_MIXED;
    f16 = cfg.precision == MIMO_PREC_FP16_MIXED;
    st = !mixed ? ST_F32 : f16 ? ST_F16 : ST_BF16;
    esz = store_bytes(st);
    {
      const int v = getenv("MIMO_FUSE_BWD_SRC") ? atoi(getenv("MIMO_FUSE_BWD_SRC")) : 1;
      fuse_bwd_src = !mixed && !cfg.inference_only && v != 0;
      fuse_bwd_pool = fuse_bwd_src && v != 3;
      fuse_bwd_head = fuse_bwd_src && v != 2;
    }
    if (cfg.precision < MIMO_PREC_FP32 || cfg.precision > MIMO_PREC_FP16_MIXED) {
      set_error("unknown precision %d", cfg.precision);
      return MIMO_ERR_INVALID;
    }
    if (cfg.norm_kind != MIMO_NORM_BATCH || cfg.act_kind != MIMO_ACT_RELU || cfg.up_kind != MIMO_UP_BILINEAR_ALIGN_CORNERS) {
      set_error("block variant norm %d / act %d / up %d is not implemented: only BatchNorm2d + ReLU + bilinear align_corners "
                "up-sampling, the reference's blocks (components.py:22-30, 77-85)", cfg.norm_kind, cfg.act_kind, cfg.up_kind);
      return MIMO_ERR_INVALID;
    }
    S = cfg.num_subnetworks;
    f = cfg.filter_base_count;
    N = cfg.batch;
    H = cfg.height;
    W = cfg.width;
    Ci = cfg.in_channels;
    Co = cfg.out_channels;
    Ci_p = round_up(Ci, 4);
    {
      // (a property of the plan, not of the stream mode: MIMO_WGRAD_STREAM=0 and the profiler's serialised pass run the same
      // launches, so the two modes stay bit-identical)
      const char* e = getenv("MIMO_WGRAD_CUS");
      const int v = e ? atoi(e) : 0;
      wg_cus = (v >= 8 && v <= 256) ? v : sched::wg_side_cus((long)N * H * W, S * f);
      const char* ea = getenv("MIMO_EVENT_ON_LAUNCH");
      ev_attach = !(ea && atoi(ea) == 0);
      const char* d = getenv("MIMO_DEBUG_WGRAD_DELAY_US");
      wg_delay_us = d ? std::max(0, std::min(atoi(d), 5000)) : 0;
    }
    if (S < 1 || f < 1 || N < 1 || Ci < 1 || Co < 2 || (Co & 1) || Co > kMaxHeadOut || f > 256) {
      set_error("unsupported configuration S=%d f=%d N=%d Ci=%d Co=%d", S, f, N, Ci, Co);
      return MIMO_ERR_INVALID;
    }
    if ((H >> 4) < 2 || (W >> 4) < 2) {
      set_error("input %dx%d too small: reflect padding needs >= 2 pixels at 1/16 resolution", H, W);
      return MIMO_ERR_INVALID;
    }
    const int H1 = H, W1 = W, H2 = H / 2, W2 = W / 2, H3 = H2 / 2, W3 = W2 / 2, H4 = H3 / 2, W4 = W3 / 2, H5 = H4 / 2,
              W5 = W4 / 2;
    std::vector<int> imgmap(Ci_p);
    for (int i = 0; i < Ci_p; ++i) imgmap[i] = i < Ci ? i : -1;
    const bool skip_alias = !(getenv("MIMO_SKIP_IN_PLACE") && atoi(getenv("MIMO_SKIP_IN_PLACE")) == 0);

    // ---- encoder (model.py:150-175) ----
    for (int s = 0; s < S; ++s) {
      DoubleConv* dc;
      MIMO_TRY(make_dc(&dc, "encoder.in_convs." + std::to_string(s) + ".double_conv", imgmap, Ci, f, f, H1, W1,
                       cfg.encoder_dropout_rate, nullptr, 0, nullptr, 0));
      MIMO_TRY(set_input(dc, IN_IMAGE, nullptr, nullptr, Ci_p, H1, W1));
      dc->subnet = s;
      enc_in.push_back(dc);
    }
    const int c2p = pad_channels(2 * f);
    x2cat.N = N;
    x2cat.H = H2;
    x2cat.W = W2;
    x2cat.C = 2 * f * S;
    x2cat.Cp = c2p * S;
    x2cat.ld = x2cat.ldda = x2cat.Cp;
    MIMO_TRY(alloc_act(&x2cat.a, (size_t)N * H2 * W2 * x2cat.Cp, st));
    if (!cfg.inference_only) MIMO_TRY(alloc_act(&x2cat.da, (size_t)N * H2 * W2 * x2cat.Cp, st));
    x2cat.chmap.assign(x2cat.Cp, -1);
    for (int s = 0; s < S; ++s)
      for (int c = 0; c < 2 * f; ++c) x2cat.chmap[s * c2p + c] = s * 2 * f + c;
    for (int s = 0; s < S; ++s) {
      DoubleConv* dc;
      MIMO_TRY(make_dc(&dc, "encoder.down1s." + std::to_string(s) + ".conv.double_conv", enc_in[s]->out.chmap, f, 2 * f,
                       2 * f, H2, W2, cfg.encoder_dropout_rate, eoff(x2cat.a, (size_t)s * c2p), x2cat.Cp,
                       x2cat.da ? eoff(x2cat.da, (size_t)s * c2p) : nullptr, x2cat.Cp));
      MIMO_TRY(set_input(dc, IN_POOL, &enc_in[s]->out, nullptr, enc_in[s]->out.Cp, H2, W2));
      dc->out.parent = &x2cat;
      dc->out.parent_choff = s * c2p;
      down1.push_back(dc);
    }
    encoder_param_floats = param_floats;  // everything registered so far belongs to the S encoders
    // ---- core (model.py:190-243) ----
    const float pc = cfg.core_dropout_rate;
    MIMO_TRY(make_dc(&down2, "core.down2.conv.double_conv", x2cat.chmap, 2 * f * S, 4 * f * S, 4 * f * S, H3, W3, pc,
                     nullptr, 0, nullptr, 0));
    MIMO_TRY(set_input(down2, IN_POOL, &x2cat, nullptr, x2cat.Cp, H3, W3));
    MIMO_TRY(make_dc(&down3, "core.down3.conv.double_conv", down2->out.chmap, 4 * f * S, 8 * f * S, 8 * f * S, H4, W4, pc,
                     nullptr, 0, nullptr, 0));
    MIMO_TRY(set_input(down3, IN_POOL, &down2->out, nullptr, down2->out.Cp, H4, W4));
    MIMO_TRY(make_dc(&down4, "core.down4.conv.double_conv", down3->out.chmap, 8 * f * S, 8 * f * S, 8 * f * S, H5, W5, pc,
                     nullptr, 0, nullptr, 0));
    MIMO_TRY(set_input(down4, IN_POOL, &down3->out, nullptr, down3->out.Cp, H5, W5));
    {
      std::vector<int> m = cat_map(down3->out, down4->out);
      MIMO_TRY(make_dc(&up1, "core.up1.conv.double_conv", m, 16 * f * S, 8 * f * S, 4 * f * S, H4, W4, pc, nullptr, 0,
                       nullptr, 0));
      MIMO_TRY(set_input(up1, IN_UPCAT, &down3->out, &down4->out, (int)m.size(), H4, W4));
      if (skip_alias) rehome_skip(down3->out, up1, {{down3, 0}});
    }
    {
      std::vector<int> m = cat_map(down2->out, up1->out);
      MIMO_TRY(make_dc(&up2, "core.up2.conv.double_conv", m, 8 * f * S, 4 * f * S, 2 * f * S, H3, W3, pc, nullptr, 0,
                       nullptr, 0));
      MIMO_TRY(set_input(up2, IN_UPCAT, &down2->out, &up1->out, (int)m.size(), H3, W3));
      if (skip_alias) rehome_skip(down2->out, up2, {{down2, 0}});
    }
    {
      std::vector<int> m = cat_map(x2cat, up2->out);
      MIMO_TRY(make_dc(&up3, "core.up3.conv.double_conv", m, 4 * f * S, 2 * f * S, f * S, H2, W2, pc, nullptr, 0, nullptr,
                       0));
      MIMO_TRY(set_input(up3, IN_UPCAT, &x2cat, &up2->out, (int)m.size(), H2, W2));
      if (skip_alias) {
        std::vector<std::pair<DoubleConv*, int>> pr;
        for (int s = 0; s < S; ++s) pr.push_back({down1[s], s * c2p});
        rehome_skip(x2cat, up3, pr);
      }
    }
    // the outputs of the core's Up blocks are read only by the next block's bilinear up-sampling: it applies their
    // BatchNorm + ReLU itself and the activated tensor is not written in a training forward (elide_output)
    elide_output(up1);
    elide_output(up2);
    elide_output(up3);
    // MaxPool2d inputs are produced by the BatchNorm + ReLU pass of the tensor they pool (one pass less per Down block)
    if (!(getenv("MIMO_POOL_FUSED") && atoi(getenv("MIMO_POOL_FUSED")) == 0)) {
      auto fuse = [this](DoubleConv* producer, DoubleConv* consumer, int choff) {
        producer->c2.pool_out = eoff(consumer->in_buf, choff);
        producer->c2.pool_ld = consumer->in_ld;
        consumer->pool_fused = true;
      };
      for (int s = 0; s < S; ++s) fuse(enc_in[s], down1[s], 0);
      for (int s = 0; s < S; ++s) fuse(down1[s], down2, s * c2p);
      fuse(down2, down3, 0);
      fuse(down3, down4, 0);
    }
    // ---- decoder (model.py:260-297) ----
    const int cin_dec = f * S + f;
    for (int s = 0; s < S; ++s) {
      std::vector<int> m = cat_map(enc_in[s]->out, up3->out);
      DoubleConv* dc;
      MIMO_TRY(make_dc(&dc, "decoder.up4s." + std::to_string(s) + ".conv.double_conv", m, cin_dec, cin_dec / 2, f, H1, W1,
                       cfg.decoder_dropout_rate, nullptr, 0, nullptr, 0));
      MIMO_TRY(set_input(dc, IN_UPCAT, &enc_in[s]->out, &up3->out, (int)m.size(), H1, W1));
      if (skip_alias) rehome_skip(enc_in[s]->out, dc, {{enc_in[s], 0}});
      // read by the 1x1 head (forward and backward) only; the element-wise final dropout would need the activated tensor
      if (cfg.final_dropout_rate <= 0.f) elide_output(dc);
      up4.push_back(dc);
    }
    for (int s = 0; s < S; ++s) {
      Head h;
      h.s = s;
      const std::string p = "decoder.outcs." + std::to_string(s) + ".conv";
      h.off_w = add_tensor(p + ".weight", {Co, f, 1, 1}, 0);
      h.off_b = add_tensor(p + ".bias", {Co}, 0);
      heads.push_back(h);
    }
    // reorder dcs into the oracle's forward-spec order: enc_in[*], down1[*], core..., up4[*]
    // (make_dc pushed them in exactly that order already).

    // ---- scratch ----
    const int fp = pad_channels(f);
    cap_partial = std::max(cap_partial, (size_t)kEwMaxBlocks * (Co * fp + Co));
    cap_sums = std::max(cap_sums, (size_t)kMaxChunks * 2 * round_up(Co * fp + Co, 64));
    if (cfg.inference_only) cap_act = cap_pad = cap_slab = 1;  // backward scratch: never touched
    MIMO_TRY(alloc_act(&s_dz, cap_act, st));
    {
      const char* we = getenv("MIMO_WGRAD_STREAM");
      // default ON (round 2): the weight gradient of layer L runs on a side stream beside the BatchNorm-backward /
      // gather kernels of layer L-1 (bandwidth-bound: they co-reside with the persistent MFMA workgroup on a CU) and
      // queues behind the data gradient of layer L; dz ping-pongs between wg_bufs buffers.  Measured +1.7 ... +4.7 %
      // images/s on three boxes (within noise on a fourth); results are bit-identical to the single-stream order.
      // With the profiler armed (bench.py's second pass) everything runs on the caller's stream.
      wg_async = !(we && atoi(we) == 0) && !cfg.inference_only;
      for (int i = 0; i < kDzBufs; ++i) s_dz2[i] = s_dz;
      if (!cfg.inference_only)
        for (int i = 0; i < kDzBufs; ++i) MIMO_TRY(dalloc(&s_dzmax2[i], kDzMaxSlots));
      if (any_mixed_dz) {
        MIMO_TRY(dalloc(&s_dzs2[0], cap_act));
        for (int i = 1; i < kDzBufs; ++i) s_dzs2[i] = s_dzs2[0];
        if (wg_async)
          for (int i = 1; i < wg_bufs; ++i) MIMO_TRY(dalloc(&s_dzs2[i], cap_act));
      }
      if (wg_async) {
        for (int i = 1; i < wg_bufs; ++i) MIMO_TRY(alloc_act(&s_dz2[i], cap_act, st));
        // LOWEST stream priority — not for the scheduling (round 4 measured no effect of the priority on the step) but for the
        // hardware queue: the runtime deals streams of one priority class round-robin onto a handful of hardware queues
        // (4 by default), and a side stream that lands on the caller's queue is silently serialised behind it — seen in
        // round 6 in bench.py's one-rank RCCL route, where the process group's streams had taken the other queues: no
        // overlap at all, 5.8 instead of 4.4 ms per step at 4 images per GPU (profiles/r06/b4/queue_collision.txt).  The
        // priority classes draw from separate queue pools, and callers run on default-priority streams.
        // MIMO_WGRAD_STREAM_PRIORITY=0 restores a default-priority stream (A/B).
        {
          int least = 0, greatest = 0;
          MIMO_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
          const char* pe = getenv("MIMO_WGRAD_STREAM_PRIORITY");
          const bool low = !(pe && atoi(pe) == 0) && least != greatest;
          if (low)
            MIMO_HIP_CHECK(hipStreamCreateWithPriority(&wg_stream, hipStreamNonBlocking, least));
          else
            MIMO_HIP_CHECK(hipStreamCreateWithFlags(&wg_stream, hipStreamNonBlocking));
        }
        for (int i = 0; i < wg_bufs; ++i)
          for (hipEvent_t* e : {&ev_dz[i], &ev_wg[i]}) MIMO_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
        MIMO_HIP_CHECK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
        MIMO_HIP_CHECK(hipEventCreateWithFlags(&ev_stage, hipEventDisableTiming));
      }
    }
    MIMO_TRY(alloc_act(&s_dxpadA, cap_pad, st));
    MIMO_TRY(alloc_act(&s_dxpadB, cap_pad, st));
    MIMO_TRY(dalloc(&s_wslab, cap_slab));
    if (cap_kpart) MIMO_TRY(dalloc(&s_kpart, cap_kpart));
    MIMO_TRY(dalloc(&s_partial, cap_partial));
    if (fuse_bwd_src) MIMO_TRY(dalloc(&s_headpart, (size_t)kBnReduceMaxBlocks * (2 * pad_channels(f) + 2)));
    MIMO_TRY(dalloc(&s_sums, cap_sums));
    MIMO_TRY(dalloc(&s_tickets, kColsumMaxGroups));
    MIMO_TRY(dalloc(&s_losspart, (size_t)S * 512));
    MIMO_TRY(dalloc(&d_status, 1));
    // ---- weight repack job tables ----
    {
      std::vector<PackJob> jobs, dg;
      for (auto& dc : dcs)
        for (ConvBN* L : {&dc->c1, &dc->c2}) {
          PackJob j{};
          j.w_off = L->off_w;
          j.bias_off = L->off_b;
          j.bias_dst = L->bias_p;
          j.bias_n = L->Cout;
          j.cout = L->Cout;
          j.cin = L->Cin;
          j.rows_pad = L->cout_pad;
          j.cols = L->cin_p;
          j.row_map = L->fwd_row_map;
          j.col_map = L->cin_map;
          j.transposed = 0;
          // split16 / fp16-mixed: fp16 (hi, lo) pairs x 2^8; bf16 / bf16-mixed: bf16 pairs (single-MFMA modes read hi only)
          j.kind = L->fwd_split ? ((cfg.precision == MIMO_PREC_BF16 || cfg.precision == MIMO_PREC_BF16_MIXED) ? 2 : 1) : 0;
          j.dst = L->fwd_split ? L->wf16 : (void*)L->wf;
          j.total = L->fwd_split ? ceil_div(j.cols, 32) * 9 * j.rows_pad * 32 : 9 * j.rows_pad * j.cols;
          j.pair = L->fwd_split ? conv3x3_pair_tail(fwd_mode(), L->cin_p, L->H, L->W) : 0;
          if (L->fwd_wide) {  // conv_wide.hip layouts: 16-channel chunks of fp16 pairs / 32-channel chunks of 16-bit values
            j.kind = !mixed ? 3 : f16 ? 5 : 6;
            j.map_rows = L->cout_pad;
            j.rows_pad = L->fwd_wide;
            j.total = mixed ? ceil_div(j.cols, 32) * 9 * j.rows_pad * 32 : ceil_div(j.cols, 16) * 9 * j.rows_pad * 16;
            j.pair = 0;
          }
          j.wmax = (j.kind == 1 || j.kind == 3 || j.kind == 5) ? L->wmax : nullptr;
          jobs.push_back(j);
          PackJob d{};
          d.w_off = L->off_w;
          d.bias_n = 0;
          d.cout = L->Cout;
          d.cin = L->Cin;
          d.rows_pad = L->dg_rows;
          d.cols = L->cout_p;
          d.row_map = L->dg_row_map;
          d.col_map = L->dg_col_map;
          d.transposed = 1;
          d.kind = L->dg_split ? (f16 ? 1 : 2) : 0;
          d.dst = L->dg_split ? L->wd16 : (void*)L->wd;
          d.total = L->dg_split ? ceil_div(d.cols, 32) * 9 * d.rows_pad * 32 : 9 * d.rows_pad * d.cols;
          d.pair = L->dg_split ? conv3x3_pair_tail(dgrad_mode(), L->cout_p, L->H + 2, L->W + 2) : 0;
          if (L->dg_wide) {  // bf16 pairs / 16-bit values
            d.kind = !mixed ? 4 : f16 ? 5 : 6;
            d.map_rows = L->dg_rows;
            d.rows_pad = L->dg_wide;
            d.total = mixed ? ceil_div(d.cols, 32) * 9 * d.rows_pad * 32 : ceil_div(d.cols, 16) * 9 * d.rows_pad * 16;
            d.pair = 0;
          }
          dg.push_back(d);
        }
      n_fwd_jobs = (int)jobs.size();
      if (!cfg.inference_only) jobs.insert(jobs.end(), dg.begin(), dg.end());
      n_all_jobs = (int)jobs.size();
      for (auto& j : jobs) pack_max_total = std::max(pack_max_total, j.total);
      MIMO_TRY(dalloc(&pack_jobs, jobs.size()));
      MIMO_HIP_CHECK(hipMemcpy(pack_jobs, jobs.data(), jobs.size() * sizeof(PackJob), hipMemcpyHostToDevice));
    }
    // hipGraph staging
    const char* ge = getenv("MIMO_HIP_GRAPH");
    graph_enabled = !(ge && atoi(ge) == 0);
    MIMO_TRY(dalloc(&g_x, (size_t)N * S * Ci * H * W));
    MIMO_TRY(dalloc(&g_out, (size_t)N * S * Co * H * W));
    MIMO_TRY(dalloc(&g_perm, (size_t)S * N));
    {
      // training-step graphs: opt-in (MIMO_TRAIN_GRAPH=1).  Built, bit-identical to eager launches, and measured NEUTRAL
      // (profiles/r06/train_graph.txt: 4.42 vs 4.42 ms per step at 4 images per GPU, 24.1 vs 23.9 at batch 32; the host spends
      // 0.76 ms in the hipGraphLaunch of the ~190-node backward graph — what the eager launches cost it), so the default
      // keeps the launches eager
      const char* te = getenv("MIMO_TRAIN_GRAPH");
      train_graph = graph_enabled && !cfg.inference_only && te && atoi(te) != 0;
      if (train_graph) {
        MIMO_TRY(dalloc(&g_label, (size_t)N * (Co / 2) * H * W));
        MIMO_TRY(dalloc(&g_lmask, (size_t)N * H * W));
        MIMO_TRY(dalloc(&g_lperm, (size_t)S * N));
        MIMO_TRY(dalloc(&g_dloss, (size_t)S));
      }
    }
    g_masks.resize(dcs.size());
    g_mask_ptrs.assign(dcs.size(), nullptr);
    for (size_t i = 0; i < dcs.size(); ++i) MIMO_TRY(dalloc(&g_masks[i], (size_t)N * dcs[i]->c2.Cout));
    {
      if (dcs.size() + 1 + S > 56) {
        set_error("too many dropout sites (%d) for the in-engine generator", (int)dcs.size() + 1 + S);
        return MIMO_ERR_INVALID;
      }
      std::vector<Dropout2dSite> sites(dcs.size());
      for (size_t i = 0; i < dcs.size(); ++i) {
        sites[i] = Dropout2dSite{g_masks[i], N * dcs[i]->c2.Cout, dcs[i]->drop_p};
        max_site_count = std::max(max_site_count, sites[i].count);
      }
      MIMO_TRY(dalloc(&d_sites, sites.size()));
      MIMO_HIP_CHECK(hipMemcpy(d_sites, sites.data(), sites.size() * sizeof(Dropout2dSite), hipMemcpyHostToDevice));
      elem_rng_on.assign(1 + S, 0);
      eff_masks.assign(dcs.size(), nullptr);
    }
    MIMO_HIP_CHECK(hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking));
    return MIMO_OK;
  }

  // ------------------------------------------------------------------ forward ------------
  // every layer's forward weight layout (+ bias copy) and, with_dgrad, the transposed data-gradient layout:
  // one launch over the job table
  int pack_all(bool with_dgrad, hipStream_t st) {
    // (fp16 forward images: the layers' max |w| first — the scale of an image follows it from |w| >= 128 up, w16_scale)
    if ((cfg.precision == MIMO_PREC_SPLIT16 || cfg.precision == MIMO_PREC_FP16_MIXED) && wmax_used > 0) {
      MIMO_HIP_CHECK(hipMemsetAsync(wmax_pool, 0, (size_t)wmax_used * sizeof(unsigned), st));
      MIMO_TRY(wabsmax_jobs_launch(pack_jobs, n_fwd_jobs, pack_max_total, params, st, d_status));
    }
    return pack_jobs_launch(pack_jobs, with_dgrad ? n_all_jobs : n_fwd_jobs, pack_max_total, params, st);
  }

  // elide: the activated tensor of this layer is not written in this call (its readers apply BatchNorm + ReLU to z)
  int convbn_forward(ConvBN& L, bool training, const float* mask, bool elide, hipStream_t st) {
    // inference: BN(eval) + ReLU (+ channel-dropout) in the conv epilogue — not for the fp32-kernel image convolution
    // of the 16-bit storage modes, whose output type differs from the activation type
    const bool fused = fwd_no_grad && !(mixed && !L.fwd_split);
    if (!training && need_derive)
      MIMO_TRY(bn_eval_prepare_launch(L.Cout, L.cout_p, params + L.off_gamma, params + L.off_beta, bnbuf + L.off_rm,
                                      bnbuf + L.off_rv, cfg.bn_eps, L.mean, L.invstd, L.scale, L.shift, st, d_status));
    // Training forward only.  (A forward without a graph folds BatchNorm + ReLU into every convolution's epilogue, and an
    // eval-mode forward WITH a graph — FGSM — keeps the separate pass, whose finiteness test feeds the numerics status word:
    // the activated tensors exist in both.  The weight gradient applies scale / shift to z either way: identical values.)
    const bool fin = L.fuse_in && training && !fwd_no_grad;
    ConvLaunch a;
    a.x = fin ? L.in_z : L.in;
    if (fin) {
      a.in_scale = L.in_scale;
      a.in_shift = L.in_shift;
    }
    a.y = fused ? L.a : L.z;  // fused: the activation type; else z (fp32 on the fp32 kernel family)
    if (fused) {
      a.ep_scale = L.scale;
      a.ep_shift = L.shift;
      a.ep_mask = mask;
      a.ep_mask_ld = L.Cout;
      a.status = d_status;
    }
    a.w = L.wf;
    a.bias = L.bias_p;
    a.stats = training ? s_partial : nullptr;
    a.N = L.N;
    a.Hi = a.Ho = L.H;
    a.Wi = a.Wo = L.W;
    a.ldx = fin ? L.ld_in_z : L.ld_in;
    a.cin_p = L.cin_p;
    a.ldy = fused ? L.ld_a : L.cout_p;
    a.cout_pad = L.cout_pad;
    a.cout_store = L.cout_p;
    a.off = 1;
    int rows = 0;
    const int64_t P = (int64_t)L.N * L.H * L.W;
    a.wpk = L.wf16;
    a.wmax = L.fwd_split ? L.wmax : nullptr;
    a.pair = (L.fwd_split && !L.fwd_wide) ? conv3x3_pair_tail(fwd_mode(), L.cin_p, L.H, L.W) : 0;
    a.wide = L.fwd_wide;
    int pr = prof_begin(MIMO_PROF_CONV_FWD, st);
    if (L.fwd_split)  // (training forward: with the K split of conv3x3_ksplit where that pays — few tiles, a long K walk)
      MIMO_TRY(conv3x3_bf16x3_launch_k(a, fwd_mode(), &rows, st, (training && !fused) ? s_kpart : nullptr, cap_kpart));
    else if (L.thin)
      MIMO_TRY(conv3x3_thin_launch(a, L.Cin, &rows, st));
    else
      MIMO_TRY(conv3x3_launch(a, &rows, st));
    prof_end(pr, 18.0 * L.Cin * L.Cout * (double)P, 4.0 * (double)P * (L.Cin + L.Cout), st);
    if (training) {
      if (rows <= kColsumMaxRows) {
        MIMO_TRY(bn_fwd_stats_launch(s_partial, rows, L.cout_pad, L.Cout, L.cout_p, P, params + L.off_gamma,
                                     params + L.off_beta, bnbuf + L.off_rm, bnbuf + L.off_rv, cfg.bn_momentum, cfg.bn_eps,
                                     L.mean, L.invstd, L.scale, L.shift, colsum(), st));
      } else {
        int chunks = 0;
        MIMO_TRY(rowsum_launch(s_partial, rows, 2 * L.cout_pad, s_sums, &chunks, st));
        MIMO_TRY(bn_fwd_finalize_launch(s_sums, chunks, L.cout_pad, L.Cout, L.cout_p, P, params + L.off_gamma,
                                        params + L.off_beta, bnbuf + L.off_rm, bnbuf + L.off_rv, cfg.bn_momentum,
                                        cfg.bn_eps, L.mean, L.invstd, L.scale, L.shift, st));
      }
    }
    if (!fused && !elide) {
      pr = prof_begin(MIMO_PROF_BN_RELU_FWD, st);
      if (L.pool_out)
        MIMO_TRY(bn_relu_pool_fwd_launch(L.z, L.dtz, L.cout_p, L.a, this->st, L.ld_a, L.scale, L.shift, mask, L.Cout, L.cout_p, L.N,
                                         L.H, L.W, L.pool_out, L.pool_ld, st, training ? nullptr : d_status));
      else
        MIMO_TRY(bn_relu_fwd_launch(L.z, L.dtz, L.cout_p, L.a, this->st, L.ld_a, L.scale, L.shift, mask, L.Cout, L.cout_p, P,
                                    L.H * L.W, st, training ? nullptr : d_status));
      prof_end(pr, 0.0, (L.pool_out ? 9.0 : 8.0) * (double)P * L.cout_p, st);
    }
    return MIMO_OK;
  }

  // Output activation of an Up block read through BatchNorm + ReLU by its consumers (Act::z): split16 training plans, no
  // Dropout2d on the block (its multipliers act on the activated tensor), MIMO_FUSE_BN_IN != 0.  The decoders' blocks
  // additionally need the element-wise final dropout off (checked where they are built).
  void elide_output(DoubleConv* dc) {
    const bool on = !(getenv("MIMO_FUSE_BN_IN") && atoi(getenv("MIMO_FUSE_BN_IN")) == 0);
    if (!on || cfg.precision != MIMO_PREC_SPLIT16 || cfg.inference_only || dc->drop_p > 0.f || dc->c2.pool_out ||
        dc->c2.dtz != ST_F32)
      return;
    dc->c2.act_elided = true;
    dc->out.z = dc->c2.z;
    dc->out.z_ld = dc->c2.cout_p;
    dc->out.z_scale = dc->c2.scale;
    dc->out.z_shift = dc->c2.shift;
  }

  // the readers of this block's output go through z in this call (Act::z_live); dc->mask must be this call's
  bool z_live_rule(const DoubleConv* dc, bool training, bool elem_mask_on_output) const {
    return dc->c2.act_elided && dc->out.z && training && !fwd_no_grad && !dc->mask && !elem_mask_on_output;
  }

  int dc_forward(DoubleConv* dc, bool training, hipStream_t st, bool elem_mask_on_output = false) {
    const int h = dc->c1.H, w = dc->c1.W;
    dc->out.z_live = z_live_rule(dc, training, elem_mask_on_output);
    const int blk = prof_begin(kProfTierBase + 2 * tier_of(h), st);
    if (dc->kind == IN_POOL) {
      Act* s = dc->src0;
      if (!(dc->pool_fused && !fwd_no_grad))  // else: already written by the producers' BatchNorm + ReLU pass
        MIMO_TRY(maxpool_fwd_launch(s->a, this->st, s->ld, N, s->H, s->W, s->Cp, dc->in_buf, dc->in_ld, st));
    } else if (dc->kind == IN_UPCAT) {
      Act *sk = dc->src0, *lo = dc->src1;
      const int pr = prof_begin(MIMO_PROF_UPCAT_FWD, st);
      const bool lz = lo->z_live;  // the low-resolution tensor through its BatchNorm + ReLU
      MIMO_TRY(upcat_fwd_launch(dc->skip_in_place ? nullptr : sk->a, this->st, sk->ld, sk->Cp, lz ? lo->z : lo->a,
                                lz ? lo->z_ld : lo->ld, lo->Cp, N, h, w, lo->H, lo->W, dc->in_buf, st, lz ? lo->z_scale : nullptr,
                                lz ? lo->z_shift : nullptr));
      // writes the up-sampled channels at (h, w), reads the low-resolution tensor once
      prof_end(pr, 0.0, 4.0 * lo->Cp * ((double)N * h * w + (double)N * lo->H * lo->W), st);
    }

    MIMO_TRY(convbn_forward(dc->c1, training, nullptr, dc->c1.act_elided && training, st));
    MIMO_TRY(convbn_forward(dc->c2, training, dc->mask, dc->out.z_live, st));
    prof_end(blk, 0.0, 0.0, st);
    return MIMO_OK;
  }

  int forward(const mimo_forward_args* args, hipStream_t st) {
    if (!params || !bnbuf) {
      set_error("mimo_forward: parameters not bound (mimo_plan_bind)");
      return MIMO_ERR_STATE;
    }
    if (!args || !args->x || !args->out) {
      set_error("mimo_forward: null argument");
      return MIMO_ERR_INVALID;
    }
    if (cfg.inference_only && (args->training || !args->no_grad)) {
      set_error("mimo_forward: inference-only plan (mimo_config.inference_only) needs training = 0 and no_grad = 1");
      return MIMO_ERR_STATE;
    }
    // ---- in-engine dropout: draw the Dropout2d multipliers of the flagged sites, note the element-wise ones ----
    mimo_forward_args a2 = *args;
    const int64_t* const orig_perm = args->perm;
    {
      const int ndc = (int)dcs.size();
      uint64_t active = 0;
      bool any_mask = false;
      for (int i = 0; i < ndc; ++i) {
        eff_masks[i] = args->drop_masks ? args->drop_masks[i] : nullptr;
        if (args->rng_sites && args->rng_sites[i] && dcs[i]->drop_p > 0.f) {
          active |= 1ull << i;
          eff_masks[i] = g_masks[i];
        }
        any_mask |= eff_masks[i] != nullptr;
      }
      rng_seed = args->rng_seed;
      rng_offset = args->rng_offset;
      bool any_elem = false;
      for (int j = 0; j <= S; ++j) {
        elem_rng_on[j] = args->rng_sites && args->rng_sites[ndc + j] && elem_rate(j) > 0.f &&
                         !(args->elem_masks && args->elem_masks[j]);
        any_elem |= elem_rng_on[j] != 0;
      }
      MIMO_TRY(dropout2d_masks_launch(d_sites, ndc, max_site_count, active, rng_seed, rng_offset, (hipStream_t)st));
      a2.drop_masks = any_mask ? eff_masks.data() : nullptr;
      a2.rng_sites = nullptr;
      if (any_elem && !a2.elem_masks) {  // keeps the call off the hipGraph paths (the generator state changes per call)
        static const float* const kNoElemMasks[64] = {};
        a2.elem_masks = kNoElemMasks;
      }
      args = &a2;
    }
    // what has to be (re)derived from the parameters in this call, and whether anything is kept for a backward
    const bool training_call = args->training != 0;
    fwd_no_grad = !training_call && args->no_grad != 0;
    need_derive = training_call || args->param_version == 0 || args->param_version != derived_version;
    const int64_t version_after = training_call ? -1 : (args->param_version != 0 ? args->param_version : -1);
    const int64_t img = (int64_t)Ci * H * W;
    const bool x5 = args->stride_s == img && args->stride_n == (int64_t)S * img;
    const bool x4 = args->stride_s == 0 && args->stride_n == img;
    const int64_t rows = args->x_rows > 0 ? args->x_rows : N;
    last_x_rows = rows;
    last_perm_arg = nullptr;
    fwd_graphed = loss_staged = false;
    // eval mode replays a graph while the packed weights stand (need_derive: the repack is not part of that graph); a
    // training forward always repacks — inside its graph
    const bool graphable = graph_enabled && !prof_on && (x5 || x4) && !args->elem_masks && rows <= N &&
                           (training_call ? (train_graph && tg_captures < kMaxTrainCaptures) : !need_derive);
    uint64_t key = 1 | (x5 ? 2 : 0) | (args->perm ? 4 : 0) | (fwd_no_grad ? 8 : 0) | (training_call ? 16 : 0);
    for (size_t i = 0; i < dcs.size(); ++i)
      if (args->drop_masks && args->drop_masks[i]) key |= 1ull << (8 + i);
    bool eager = !graphable;
    if (graphable && training_call && !(tg_fwd && key == tg_fwd_key) && key != tg_fwd_seen) {
      tg_fwd_seen = key;  // first sighting of this call shape: eager (captured when it comes again)
      eager = true;
    }
    if (eager) {
      const int rc = forward_impl(args, st);
      derived_version = rc == MIMO_OK ? version_after : -1;
      return rc;
    }
    // ---- stage the caller's tensors, (re)capture if the call shape changed, replay ----
    MIMO_HIP_CHECK(hipMemcpyAsync(g_x, args->x, (size_t)rows * (x5 ? S : 1) * img * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (args->perm)
      MIMO_HIP_CHECK(hipMemcpyAsync(g_perm, args->perm, (size_t)S * N * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
    for (size_t i = 0; i < dcs.size(); ++i) {
      const bool on = args->drop_masks && args->drop_masks[i];
      g_mask_ptrs[i] = on ? g_masks[i] : nullptr;
      if (on && args->drop_masks[i] != g_masks[i])  // (an in-engine site was drawn straight into the staging buffer)
        MIMO_HIP_CHECK(hipMemcpyAsync(g_masks[i], args->drop_masks[i], (size_t)N * dcs[i]->c2.Cout * sizeof(float),
                                      hipMemcpyDeviceToDevice, st));
    }
    hipGraphExec_t* exec = training_call ? &tg_fwd : &graph_exec;
    uint64_t* ekey = training_call ? &tg_fwd_key : &graph_key;
    if (!*exec || key != *ekey) {
      if (*exec) {
        (void)hipGraphExecDestroy(*exec);
        *exec = nullptr;
      }
      mimo_forward_args ga = *args;
      ga.x = g_x;
      ga.perm = args->perm ? g_perm : nullptr;
      ga.drop_masks = args->drop_masks ? g_mask_ptrs.data() : nullptr;
      ga.out = g_out;
      MIMO_TRY(capture([&](hipStream_t cs) { return forward_impl(&ga, cs); }, exec));
      *ekey = key;
      if (training_call) ++tg_captures;
    }
    elem_masks.clear();
    MIMO_HIP_CHECK(hipGraphLaunch(*exec, st));
    MIMO_HIP_CHECK(hipMemcpyAsync(args->out, g_out, (size_t)N * S * Co * H * W * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (training_call) {
      // the per-call state forward_impl / dc_forward leave behind for the loss and the backward, on the STAGED tensors (a
      // backward graph must not see the caller's pointers): masks, which activated tensors were elided, the logits
      elem_masks.assign(1 + S, nullptr);
      for (size_t i = 0; i < dcs.size(); ++i) {
        dcs[i]->mask = g_mask_ptrs[i];
        dcs[i]->out.z_live = z_live_rule(dcs[i].get(), true, false);
      }
      out = g_out;
      fwd_graphed = true;
      last_perm_arg = orig_perm;
    } else {
      for (size_t i = 0; i < dcs.size(); ++i) {
        dcs[i]->mask = args->drop_masks ? args->drop_masks[i] : nullptr;
        dcs[i]->out.z_live = false;  // (an eval-mode forward writes every activated tensor)
      }
      out = args->out;
    }
    fwd_done = true;
    fwd_training = training_call;
    had_perm = args->perm != nullptr;
    loss_done = false;
    derived_version = version_after;  // -1 after a training forward: the optimiser step invalidates the packed weights
    return MIMO_OK;
  }

  // run `body` under stream capture on cap_stream and instantiate the resulting graph
  template <typename F>
  int capture(F body, hipGraphExec_t* exec) {
    MIMO_HIP_CHECK(hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal));
    capturing = true;
    const int rc = body(cap_stream);
    capturing = false;
    hipGraph_t graph = nullptr;
    const hipError_t ce = hipStreamEndCapture(cap_stream, &graph);
    if (rc != MIMO_OK) {
      if (graph) (void)hipGraphDestroy(graph);
      return rc;
    }
    MIMO_HIP_CHECK(ce);
    const hipError_t ie = hipGraphInstantiate(exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    MIMO_HIP_CHECK(ie);
    return MIMO_OK;
  }

  int forward_impl(const mimo_forward_args* args, hipStream_t st) {
    if (!params || !bnbuf) {
      set_error("mimo_forward: parameters not bound (mimo_plan_bind)");
      return MIMO_ERR_STATE;
    }
    if (!args || !args->x || !args->out) {
      set_error("mimo_forward: null argument");
      return MIMO_ERR_INVALID;
    }
    const bool training = args->training != 0;
    for (size_t i = 0; i < dcs.size(); ++i) dcs[i]->mask = args->drop_masks ? args->drop_masks[i] : nullptr;
    elem_masks.assign(1 + S, nullptr);
    if (args->elem_masks)
      for (int i = 0; i <= S; ++i) elem_masks[i] = args->elem_masks[i];
    if (need_derive) MIMO_TRY(pack_all(training, st));
    for (int s = 0; s < S; ++s)
      MIMO_TRY(pack_input_launch(args->x, args->stride_n, args->stride_s, args->perm, s, N, Ci, H, W, enc_in[s]->in_buf,
                                 Ci_p, st));
    for (int s = 0; s < S; ++s) {  // chain by chain
      MIMO_TRY(dc_forward(enc_in[s], training, st));
      MIMO_TRY(dc_forward(down1[s], training, st));
    }
    MIMO_TRY(dc_forward(down2, training, st));
    MIMO_TRY(dc_forward(down3, training, st));
    MIMO_TRY(dc_forward(down4, training, st));
    // center_dropout: in place on down4's output, whose only reader is up1's upsample (the BN/ReLU
    // backward recomputes its mask from z, not from a)
    if (elem_masks[0] || elem_rng_on[0]) {
      const ElemRng g = elem_rng(0);
      MIMO_TRY(elem_mask_mul_launch(down4->out.a, this->st, down4->out.ld, elem_masks[0], N, down4->out.C, down4->out.Cp,
                                    down4->out.H * down4->out.W, st, elem_masks[0] ? nullptr : &g));
    }
    MIMO_TRY(dc_forward(up1, training, st));
    MIMO_TRY(dc_forward(up2, training, st));
    MIMO_TRY(dc_forward(up3, training, st));
    for (int s = 0; s < S; ++s) {
      // (an element-wise final dropout on a block whose output is elided by construction: this call materialises it)
      MIMO_TRY(dc_forward(up4[s], training, st, elem_masks[1 + s] || elem_rng_on[1 + s]));
      const Act& o = up4[s]->out;
      // final_dropouts[s]: in place, the head (forward and weight gradient) is the only reader
      if (elem_masks[1 + s] || elem_rng_on[1 + s]) {
        const ElemRng g = elem_rng(1 + s);
        MIMO_TRY(elem_mask_mul_launch(o.a, this->st, o.ld, elem_masks[1 + s], N, o.C, o.Cp, H * W, st,
                                      elem_masks[1 + s] ? nullptr : &g));
      }
      const int blk = prof_begin(kProfTierBase, st);
      const int pr = prof_begin(MIMO_PROF_HEAD_FWD, st);
      const bool oz = o.z_live;
      MIMO_TRY(head_fwd_launch(oz ? o.z : o.a, this->st, oz ? o.z_ld : o.ld, params + heads[s].off_w, params + heads[s].off_b, f, Co,
                               N, S, s, H * W, args->out, st, d_status, oz ? o.z_scale : nullptr, oz ? o.z_shift : nullptr));
      prof_end(pr, 0.0, 4.0 * (double)N * H * W * (pad_channels(f) + Co), st);
      prof_end(blk, 0.0, 0.0, st);
    }
    out = args->out;
    fwd_done = true;
    fwd_training = training;
    had_perm = args->perm != nullptr;
    loss_done = false;
    return MIMO_OK;
  }

  int loss_forward(const float* label_, const float* mask_, const int64_t* perm_, float* loss_out, hipStream_t st) {
    if (!fwd_done) {
      set_error("mimo_loss_forward: call mimo_forward first");
      return MIMO_ERR_STATE;
    }
    if (!label_ || !loss_out) {
      set_error("mimo_loss_forward: null argument");
      return MIMO_ERR_INVALID;
    }
    loss_staged = false;
    if (fwd_graphed && train_graph && last_x_rows >= 1 && last_x_rows <= N) {
      // a backward graph reads label / mask / perm: plan-owned copies (the batch's tensors have last_x_rows rows)
      const size_t hw = (size_t)H * W;
      MIMO_HIP_CHECK(hipMemcpyAsync(g_label, label_, (size_t)last_x_rows * (Co / 2) * hw * sizeof(float), hipMemcpyDeviceToDevice, st));
      label_ = g_label;
      if (mask_) {
        MIMO_HIP_CHECK(hipMemcpyAsync(g_lmask, mask_, (size_t)last_x_rows * hw * sizeof(float), hipMemcpyDeviceToDevice, st));
        mask_ = g_lmask;
      }
      if (perm_ && perm_ == last_perm_arg) {
        perm_ = g_perm;  // the forward staged this very tensor
      } else if (perm_) {
        MIMO_HIP_CHECK(hipMemcpyAsync(g_lperm, perm_, (size_t)S * N * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
        perm_ = g_lperm;
      }
      loss_staged = true;
    }
    int blocks = 0;
    MIMO_TRY(loss_fwd_launch(out, label_, mask_, perm_, N, S, Co, H * W, cfg.loss_kind, cfg.eps_min, cfg.eps_max,
                             s_losspart, &blocks, st));
    MIMO_TRY(loss_finalize_launch(s_losspart, S, blocks, (double)N * (Co / 2) * H * W, loss_out, st));
    label = label_;
    lmask = mask_;
    lperm = perm_;
    loss_done = true;
    return MIMO_OK;
  }

  // ------------------------------------------------------------------ backward -----------
  // gradient arriving at an activation: first writer assigns, later writers accumulate
  static int acc_flag(Act* t) { return t->grad_writes++ > 0 ? 1 : 0; }

  // `src`: where the gradient arriving at L's activation comes from (GradSrc, elementwise.h)
  int convbn_backward(ConvBN& L, const GradSrc& src, const float* mask, bool need_dgrad, float* dxpad_out, hipStream_t st) {
    int rows = 0;
    const int64_t P = (int64_t)L.N * L.H * L.W;
    // algorithmic bytes of the source per pixel and channel (z and dz are counted below): the pooled gradient is a quarter
    // of the image, the head's logits / labels a few floats per pixel
    const double src_b = src.kind == GS_POOL ? (1.0 + (src.skip ? 4.0 : 0.0)) : src.kind == GS_HEAD ? 0.5 : 4.0;
    int pr = prof_begin(MIMO_PROF_BN_BWD_REDUCE, st);
    MIMO_TRY(bnrelu_bwd_reduce_launch(src, this->st, L.z, L.dtz, L.cout_p, L.scale, L.shift, L.mean, L.invstd,
                                      mask, L.Cout, L.cout_p, L.N, L.H, L.W, s_partial, &rows, st));
    prof_end(pr, 0.0, (4.0 + src_b) * (double)P * L.cout_p, st);
    if (src.kind == GS_HEAD) head_rows = rows;
    MIMO_TRY(bn_bwd_stats_launch(s_partial, rows, L.Cout, L.cout_p, P, fwd_training ? 1 : 0, L.c1, L.c2,
                                 grads + L.off_gamma, grads + L.off_beta, fwd_training ? grads + L.off_b : nullptr, colsum(), st));
    // (with the profiler armed everything runs on the caller's stream: per-kernel times, not overlapped times)
    const bool async = wg_async && !prof_on;
    const int b = dz_idx;
    float* dz = s_dz2[b];
    if (async) {
      dz_idx = (dz_idx + 1) % wg_bufs;
      // last reader of this dz buffer (and, released in pairs, of the one after it)
      if (kDzBufs >= 4 && (kDzBufs & 1) == 0) {
        if ((b & 1) == 0) {
          const int w = wg_pending[b + 1] ? b + 1 : b;
          if (wg_pending[w]) MIMO_HIP_CHECK(hipStreamWaitEvent(st, ev_wg[w], 0));
          wg_pending[b] = wg_pending[b + 1] = false;  // (their last readers are behind this wait)
        } else if (wg_pending[b]) {  // not covered by the pair's wait: cannot happen in cyclic order, kept as the safe path
          MIMO_HIP_CHECK(hipStreamWaitEvent(st, ev_wg[b], 0));
        }
      } else if (wg_pending[b]) {
        MIMO_HIP_CHECK(hipStreamWaitEvent(st, ev_wg[b], 0));
      }
    }
    // the image convolution's weight gradient on the plain-FMA kernel reads dz as fp32 (no data gradient wanted: nothing
    // else reads this dz)
    const bool thin_wg = L.thin && !mixed && !need_dgrad && wgrad_thin_ok(L.Cin, L.cout_p, L.N, L.H, L.W);
    pr = prof_begin(MIMO_PROF_BN_BWD_APPLY, st);
    int dzmax_n = 0;  // per-workgroup maxima of |dz| that launch leaves (two-MFMA weight gradient)
    // "dz exists" travels with the launch that writes it (a stop event on the kernel, no hipEventRecord behind it) whenever that
    // launch is the last writer — not when a split copy of dz follows — and not under stream capture
    const bool needs_split_copy = L.wg_split && !L.dg_split && !thin_wg;
    const bool ev_on_launch = async && ev_attach && !capturing && !needs_split_copy;
    MIMO_TRY(bn_bwd_apply_launch(src, this->st, L.z, L.dtz, L.cout_p, L.scale, L.shift, L.mean, L.invstd, mask,
                                 L.Cout, L.c1, L.c2, L.cout_p, L.N, L.H, L.W, dz, (L.dg_split && !mixed && !thin_wg) ? 1 : 0,
                                 fwd_training ? nullptr : s_partial, &rows, st, (L.wg_np2 && !thin_wg) ? s_dzmax2[b] : nullptr,
                                 &dzmax_n, ev_on_launch ? ev_dz[b] : nullptr));
    prof_end(pr, 0.0, (8.0 + src_b) * (double)P * L.cout_p, st);
    // dz storage: bf16 hi|lo pairs when the data-gradient kernel is the bf16-pair one (then the weight
    // gradient is too); fp32 otherwise, with a split copy for a bf16-pair weight gradient
    const float* dz_wg = dz;
    if (needs_split_copy) {
      float* dzs = s_dzs2[b];
      MIMO_TRY(split_pairs_launch(dz, dzs, P, L.cout_p, st));
      dz_wg = dzs;
    }
    // wgrad(L) may start as soon as dz exists, next to dgrad(L)
    if (async && !ev_on_launch) MIMO_HIP_CHECK(hipEventRecord(ev_dz[b], st));
    // conv bias gradient: exactly zero in front of a training-mode BatchNorm (written by bn_bwd_stats above);
    // a real column sum of dz only after an eval-mode forward (running statistics: dz = scale * dy)
    if (!fwd_training) MIMO_TRY(colsum_vec_launch(s_partial, rows, L.cout_p, L.Cout, grads + L.off_b, colsum(), st));
    if (need_dgrad) {
      ConvLaunch a;
      a.x = dz;
      a.y = dxpad_out;
      a.w = L.wd;
      a.bias = nullptr;
      a.stats = nullptr;
      a.N = L.N;
      a.Hi = L.H;
      a.Wi = L.W;
      a.ldx = L.cout_p;
      a.cin_p = L.cout_p;
      a.Ho = L.H + 2;
      a.Wo = L.W + 2;
      a.ldy = L.cin_p;
      a.cout_pad = L.dg_rows;
      a.cout_store = L.cin_p;
      a.off = 2;
      a.wpk = L.wd16;
      a.pair = (L.dg_split && !L.dg_wide) ? conv3x3_pair_tail(dgrad_mode(), L.cout_p, L.H + 2, L.W + 2) : 0;
      a.wide = L.dg_wide;
      pr = prof_begin(MIMO_PROF_CONV_DGRAD, st);
      if (L.dg_split)
        MIMO_TRY(conv3x3_bf16x3_launch_k(a, dgrad_mode(), nullptr, st, s_kpart, cap_kpart));
      else
        MIMO_TRY(conv3x3_launch(a, nullptr, st));
      prof_end(pr, 18.0 * L.Cin * L.Cout * (double)P, 4.0 * (double)P * (L.Cin + L.Cout), st);
    }
    hipStream_t ws = st;
    if (async) {
      MIMO_HIP_CHECK(hipStreamWaitEvent(wg_stream, ev_dz[b], 0));  // (a wait refers to the record made just above)
      ws = wg_stream;
    }
    WgradLaunch wg;
    wg.x = L.fuse_in ? L.in_z : L.in;
    if (L.fuse_in) {
      wg.in_scale = L.in_scale;
      wg.in_shift = L.in_shift;
    }
    wg.dz = dz_wg;
    wg.partial = s_wslab;
    wg.N = L.N;
    wg.H = L.H;
    wg.W = L.W;
    wg.ldx = L.fuse_in ? L.ld_in_z : L.ld_in;
    wg.lddz = L.cout_p;
    wg.cin_p = L.cin_p;
    wg.cout_p = L.cout_p;
    wg.cin_pad = L.wg_cin_pad;
    wg.cout_pad = L.wg_cout_pad;
    wg.splits = L.wg_splits;
    wg.np = (cfg.precision == MIMO_PREC_BF16 || mixed) ? 1 : 3;
    if (L.wg_np2 && wg.np == 3 && !thin_wg) {  // two fp16 MFMAs per product (wgrad_split.hip NP == 2)
      wg.np = 2;
      wg.dz_absmax = s_dzmax2[b];  // (travels with the dz buffer it describes: same ping-pong index, same events)
      wg.dz_absmax_n = dzmax_n;
    }
    // 16-bit storage: activations and dz plain NHWC 16-bit; the image convolution's input stays fp32
    wg.store = !mixed ? 0 : (L.fwd_split ? (f16 ? 2 : 1) : (f16 ? 4 : 3));
    if (wg_delay_us > 0) MIMO_TRY(debug_delay_launch(wg_delay_us, ws));  // test hook: a late consumer
    pr = prof_begin(MIMO_PROF_CONV_WGRAD, ws);
    if (thin_wg)
      MIMO_TRY(wgrad_thin_launch(L.in, L.ld_in, dz, L.cout_p, L.N, L.H, L.W, L.Cin, L.Cout, L.cout_p, s_wslab, grads + L.off_w, ws));
    else if (L.wg_split)
      MIMO_TRY(wgrad_split_launch(wg, ws));
    else
      MIMO_TRY(wgrad_launch(wg, ws));
    prof_end(pr, 18.0 * L.Cin * L.Cout * (double)P, 4.0 * (double)P * (L.Cin + L.Cout), ws);
    const bool wg_ev_on_launch = async && ev_attach && !capturing && !thin_wg;
    if (!thin_wg)  // (the plain-FMA kernel's launch reduces its own partials)
      MIMO_TRY(wgrad_reduce_launch(s_wslab, L.wg_splits, L.wg_cin_pad, L.wg_cout_pad, L.cin_map, L.cin_p, L.Cin, L.Cout,
                                   grads + L.off_w, ws, wg.dz_absmax, wg.dz_absmax_n, wg_ev_on_launch ? ev_wg[b] : nullptr));
    if (async) {
      // dz buffer b AND its max |dz| slots are free again — recorded behind the REDUCTION: it reads the slots too (to take the
      // two-MFMA kernel's scale out again), and the BatchNorm backward of the layer after next overwrites them.  (Until the
      // end of round 5 the event sat in front of the reduction: a race that showed as a weight gradient off by > 1e-4 of
      // its scale in one small-geometry test when that test ran alone.)
      if (!wg_ev_on_launch) MIMO_HIP_CHECK(hipEventRecord(ev_wg[b], wg_stream));
      wg_pending[b] = true;
    }
    return MIMO_OK;
  }

  // the caller's stream waits for every weight gradient issued so far
  int wg_join(hipStream_t st) {
    bool any = false;
    for (bool p : wg_pending) any |= p;
    if (!wg_async || !any) return MIMO_OK;
    MIMO_HIP_CHECK(hipEventRecord(ev_join, wg_stream));
    MIMO_HIP_CHECK(hipStreamWaitEvent(st, ev_join, 0));
    for (bool& p : wg_pending) p = false;
    return MIMO_OK;
  }

  int dc_backward(DoubleConv* dc, bool need_input_grad, hipStream_t st, const GradSrc* head_src = nullptr) {
    const int blk = prof_begin(kProfTierBase + 2 * tier_of(dc->c1.H) + 1, st);
    const int rc = dc_backward_impl(dc, need_input_grad, st, head_src);
    prof_end(blk, 0.0, 0.0, st);
    return rc;
  }

  int dc_backward_impl(DoubleConv* dc, bool need_input_grad, hipStream_t st, const GradSrc* head_src = nullptr) {
    GradSrc src = GradSrc::plain(dc->out.da, dc->out.ldda);
    {
      const Act* root = dc->out.parent ? dc->out.parent : &dc->out;
      if (head_src) {
        src = *head_src;
      } else if (root->poolgrad) {  // pooled by a Down block that left its gradient in place (below)
        src.kind = GS_POOL;
        src.dxpad = root->poolgrad;
        src.ldp = root->poolgrad_ld;
        src.skip = root->skipgrad;
        src.ldsk = root->skipgrad_ld;
        src.choff = src.skoff = dc->out.parent ? dc->out.parent_choff : 0;
      }
    }
    MIMO_TRY(convbn_backward(dc->c2, src, dc->mask, true, s_dxpadA, st));
    float* dxB = dc->dxpad_own ? dc->dxpad_own : s_dxpadB;
    MIMO_TRY(convbn_backward(dc->c1, GradSrc::fold(s_dxpadA, dc->c1.cout_p), nullptr, need_input_grad, dxB, st));
    if (!need_input_grad) return MIMO_OK;
    const int h = dc->c1.H, w = dc->c1.W, ldp = dc->c1.cin_p;
    if (dc->kind == IN_POOL && fuse_bwd_pool && dc->dxpad_own && dc->src0->grad_writes == 0) {
      // nobody has written the pooled tensor's gradient buffer (its skip gradient, if any, waits in the Up block's own
      // buffer): the producers' BatchNorm backward reads pool route + skip fold straight from the two padded-domain buffers
      Act* s = dc->src0;
      s->poolgrad = dxB;
      s->poolgrad_ld = ldp;
    } else if (dc->kind == IN_POOL) {
      Act* s = dc->src0;
      const int pr = prof_begin(MIMO_PROF_POOL_BWD, st);
      const bool acc = acc_flag(s) != 0;
      MIMO_TRY(pool_bwd_launch(dxB, this->st, ldp, 0, s->a, s->ld, s->da, s->ldda, N, s->H, s->W, s->Cp, acc ? 1 : 0, st, s->skipgrad,
                               s->skipgrad_ld));
      // reads the pooled gradient (1/4), the activation, [the skip gradient], [the old gradient]; writes the gradient
      prof_end(pr, 0.0, 4.0 * s->Cp * (double)N * s->H * s->W * (2.25 + (s->skipgrad ? 1.0 : 0.0) + (acc ? 1.0 : 0.0)), st);
      s->skipgrad = nullptr;
    } else if (dc->kind == IN_UPCAT) {
      Act *sk = dc->src0, *lo = dc->src1;
      if (dc->dxpad_own) {  // the skip slice stays where the data gradient wrote it; the pool backward of sk folds it in
        sk->skipgrad = dxB;
        sk->skipgrad_ld = ldp;
      } else {
        MIMO_TRY(fold_slice_launch(dxB, this->st, ldp, 0, sk->da, sk->ldda, N, h, w, sk->Cp, acc_flag(sk), st));
      }
      const int pr = prof_begin(MIMO_PROF_UP_BWD, st);
      MIMO_TRY(up_bwd_launch(dxB, this->st, ldp, sk->Cp, lo->da, lo->ldda, N, h, w, lo->H, lo->W, lo->Cp, acc_flag(lo), st));
      // reads the up-sampled slice of the padded-domain gradient once, writes the low-resolution gradient
      prof_end(pr, 0.0, 4.0 * lo->Cp * ((double)N * (h + 2) * (w + 2) + (double)N * lo->H * lo->W), st);
    }
    return MIMO_OK;
  }

  // The backward in kBwdStages stages, in execution order; each stage's parameters are one contiguous range of the
  // flat gradient buffer (laid out encoder | core down2..up3 | decoder | heads), final when the stage returns, so a
  // data-parallel caller can start that range's all-reduce while the later stages run:
  //   0 heads + decoders (up4)   1 up3   2 up2   3 up1   4 down4   5 down3   6 down2   7 encoders (+ dx)
  static constexpr int kBwdStages = 8;
  void stage_range(int stage, int64_t* b, int64_t* e) const {
    DoubleConv* core[6] = {up3, up2, up1, down4, down3, down2};
    if (stage == 0) {
      *b = up4[0]->p_begin;
      *e = param_floats;
    } else if (stage == kBwdStages - 1) {
      *b = 0;
      *e = encoder_param_floats;
    } else {
      *b = core[stage - 1]->p_begin;
      *e = core[stage - 1]->p_end;
    }
  }

  // ready != nullptr (mimo_backward_stage_async): the stage's gradient range need only be final on the stream returned in
  // *ready — the side stream when the weight gradients run there (it first waits for the caller's stream: BatchNorm and head
  // gradients are written there) — and the caller's stream is NOT made to wait for the side stream: a data-parallel caller
  // issues the range's collective on *ready, and the main stream goes straight on with the next stage.  The last stage joins.
  int backward(const float* dout, const float* dloss, float* dx, int stage_first, int stage_last, hipStream_t st,
               hipStream_t* ready = nullptr) {
    if (ready) *ready = st;
    if (!fwd_done) {
      set_error("mimo_backward: call mimo_forward first");
      return MIMO_ERR_STATE;
    }
    if (!grads) {
      set_error("mimo_backward: gradient buffer not bound");
      return MIMO_ERR_STATE;
    }
    if (fwd_no_grad) {
      set_error("mimo_backward: the last forward ran with no_grad (inference epilogue, nothing saved)");
      return MIMO_ERR_STATE;
    }
    if (!dout && !dloss) {
      set_error("mimo_backward: need dout and/or dloss");
      return MIMO_ERR_INVALID;
    }
    if (dloss && !loss_done) {
      set_error("mimo_backward: dloss given but mimo_loss_forward was not called");
      return MIMO_ERR_STATE;
    }
    if (dx && had_perm) {
      set_error("mimo_backward: dx requires a forward without perm");
      return MIMO_ERR_INVALID;
    }
    if (stage_first < 0 || stage_last >= kBwdStages || stage_first > stage_last) {
      set_error("mimo_backward: bad stage range %d..%d", stage_first, stage_last);
      return MIMO_ERR_INVALID;
    }
    if (stage_first != 0 && stage_first != bwd_next_stage) {
      set_error("mimo_backward: stage %d requested, stage %d is next", stage_first, bwd_next_stage);
      return MIMO_ERR_STATE;
    }
    const bool whole = stage_first == 0 && stage_last == kBwdStages - 1;
    if (stage_first == 0) {
      // ---- training-step graphs: decide for this backward, stage dloss, capture on the second sighting of the shape ----
      tg_bwd_live = false;
      const bool single = stage_first == stage_last;
      // (the asynchronous stages — `ready` — are eager launches: a per-stage graph has to close its fork to the side stream
      // before it ends, which is the join that route exists to avoid)
      if (train_graph && !ready && fwd_graphed && loss_staged && fwd_training && !prof_on && !dout && !dx && dloss && (whole || single) &&
          tg_captures < kMaxTrainCaptures) {
        const uint64_t key = (tg_fwd_key << 3) | (lmask ? 4 : 0) | (lperm ? 2 : 0) | (whole ? 1 : 0);
        bool ready = tg_bwd_key == key && (whole ? tg_bwd[kBwdStages] != nullptr : tg_bwd[0] != nullptr);
        if (!ready && tg_bwd_seen == key) {
          for (auto& e : tg_bwd) {
            if (e) (void)hipGraphExecDestroy(e);
            e = nullptr;
          }
          tg_bwd_key = 0;
          // (a capture executes nothing: the stages can be captured one after the other, each on the host state — gradient
          // routing of the activation tensors — the stage before it left)
          if (whole) {
            MIMO_TRY(capture([&](hipStream_t cs) {
              dz_idx = 0;
              for (int stage = 0; stage < kBwdStages; ++stage) MIMO_TRY(backward_stage(stage, nullptr, g_dloss, nullptr, cs));
              return wg_join(cs);
            }, &tg_bwd[kBwdStages]));
          } else {
            for (int stage = 0; stage < kBwdStages; ++stage)
              MIMO_TRY(capture([&](hipStream_t cs) {
                dz_idx = 0;
                MIMO_TRY(backward_stage(stage, nullptr, g_dloss, nullptr, cs));
                return wg_join(cs);
              }, &tg_bwd[stage]));
          }
          tg_bwd_key = key;
          ++tg_captures;
          ready = true;
        }
        tg_bwd_seen = key;
        tg_bwd_live = ready;
        if (ready) MIMO_HIP_CHECK(hipMemcpyAsync(g_dloss, dloss, (size_t)S * sizeof(float), hipMemcpyDeviceToDevice, st));
      }
    }
    if (tg_bwd_live && !dout && !dx && (whole || stage_first == stage_last)) {
      hipGraphExec_t e = whole ? tg_bwd[kBwdStages] : tg_bwd[stage_first];
      if (!e) {
        set_error("mimo_backward: no graph for stage range %d..%d of the backward in progress", stage_first, stage_last);
        return MIMO_ERR_STATE;
      }
      MIMO_HIP_CHECK(hipGraphLaunch(e, st));
      bwd_next_stage = stage_last + 1 < kBwdStages ? stage_last + 1 : 0;
      return MIMO_OK;  // (every backward graph ends with the join of the side stream)
    }
    if (tg_bwd_live) {
      set_error("mimo_backward: stage %d..%d differs from what stage 0 of this backward was called with", stage_first, stage_last);
      return MIMO_ERR_STATE;
    }
    for (int stage = stage_first; stage <= stage_last; ++stage) MIMO_TRY(backward_stage(stage, dout, dloss, dx, st));
    bwd_next_stage = stage_last + 1 < kBwdStages ? stage_last + 1 : 0;
    if (ready && wg_async && !prof_on && stage_last < kBwdStages - 1) {
      MIMO_HIP_CHECK(hipEventRecord(ev_stage, st));
      MIMO_HIP_CHECK(hipStreamWaitEvent(wg_stream, ev_stage, 0));
      *ready = wg_stream;
      return MIMO_OK;
    }
    return wg_join(st);  // the gradients of the stages run so far are final for the caller (all-reduce)
  }

  int backward_stage(int stage, const float* dout, const float* dloss, float* dx, hipStream_t st) {
    switch (stage) {
      case 0: {
        for (auto& dc : dcs) {
          dc->out.grad_writes = 0;
          dc->out.skipgrad = nullptr;
          dc->out.poolgrad = nullptr;
        }
        x2cat.grad_writes = 0;
        x2cat.skipgrad = nullptr;
        x2cat.poolgrad = nullptr;
        if (!fwd_training) {  // eval-mode forward skipped the dgrad weight packing
          MIMO_TRY(pack_all(true, st));
        }
        const int fp = pad_channels(f);
        for (int s = S - 1; s >= 0; --s) {
          DoubleConv* dc = up4[s];
          int rows = 0;
          const float* em = elem_masks.empty() ? nullptr : elem_masks[1 + s];
          if (fuse_bwd_head && Co == 2 && !dc->mask && !em && !elem_rng_on[1 + s]) {
            // the head's input gradient is formed by the BatchNorm backward of the decoder's last convolution itself (GS_HEAD):
            // head_bwd is not launched, the gradient of the decoder output is never written
            GradSrc hs;
            hs.kind = GS_HEAD;
            hs.head = HeadGrad{params + heads[s].off_w, f, Co, N, S, s, H * W, out, dout, dloss, label, lmask, lperm, cfg.loss_kind,
                               cfg.eps_min, cfg.eps_max, 1.f / (float)((double)N * (Co / 2) * H * W), s_headpart};
            MIMO_TRY(dc_backward(dc, true, st, &hs));
            // (the head's partial rows: one per workgroup of that reduction)
            MIMO_TRY(head_bwd_stats_launch(s_headpart, head_rows, f, fp, Co, grads + heads[s].off_w, grads + heads[s].off_b, colsum(), st));
            continue;
          }
          const int blk = prof_begin(kProfTierBase + 1, st);
          const int pr = prof_begin(MIMO_PROF_HEAD_BWD, st);
          // (the tensor the forward's head read: z through scale / shift + ReLU, or the materialised — possibly masked — one)
          const Act& o = dc->out;
          const bool oz = o.z_live;
          MIMO_TRY(head_bwd_launch(oz ? o.z : o.a, this->st, oz ? o.z_ld : o.ld, params + heads[s].off_w, f, fp, Co, N, S, s, H * W,
                                   out, dout, dloss, label, lmask, lperm, cfg.loss_kind, cfg.eps_min, cfg.eps_max, dc->out.da,
                                   s_partial, &rows, st, oz ? o.z_scale : nullptr, oz ? o.z_shift : nullptr));
          prof_end(pr, 0.0, 4.0 * (double)N * H * W * (2.0 * fp + Co + Co / 2), st);
          MIMO_TRY(head_bwd_stats_launch(s_partial, rows, f, fp, Co, grads + heads[s].off_w, grads + heads[s].off_b, colsum(), st));
          if (em || elem_rng_on[1 + s]) {
            const ElemRng g = elem_rng(1 + s);
            MIMO_TRY(elem_mask_mul_launch(dc->out.da, this->st, dc->out.ldda, em, N, dc->out.C, dc->out.Cp, H * W, st,
                                          em ? nullptr : &g));
          }
          prof_end(blk, 0.0, 0.0, st);
          MIMO_TRY(dc_backward(dc, true, st));
        }
        return MIMO_OK;
      }
      case 1: return dc_backward(up3, true, st);
      case 2: return dc_backward(up2, true, st);
      case 3: return dc_backward(up1, true, st);
      case 4:
        {
          const float* em = elem_masks.empty() ? nullptr : elem_masks[0];
          if (em || elem_rng_on[0]) {
            const ElemRng g = elem_rng(0);
            MIMO_TRY(elem_mask_mul_launch(down4->out.da, this->st, down4->out.ldda, em, N, down4->out.C, down4->out.Cp,
                                          down4->out.H * down4->out.W, st, em ? nullptr : &g));
          }
        }
        return dc_backward(down4, true, st);
      case 5: return dc_backward(down3, true, st);
      case 6: return dc_backward(down2, true, st);
      default: return backward_encoders(dx, st);
    }
  }

  int backward_encoders(float* dx, hipStream_t st) {
    for (int s = S - 1; s >= 0; --s) MIMO_TRY(dc_backward(down1[s], true, st));
    for (int s = S - 1; s >= 0; --s) {
      MIMO_TRY(dc_backward(enc_in[s], dx != nullptr, st));
      if (dx) MIMO_TRY(unpack_dx_launch(s_dxpadB, this->st, Ci_p, N, S, s, Ci, H, W, dx, st));
    }
    return MIMO_OK;
  }
};

// ======================================================================= C ABI ==========
extern "C" {

const char* mimo_last_error(void) { return mimo::last_error(); }
int mimo_version(void) { return 1; }

int mimo_plan_create(const mimo_config* cfg, mimo_plan** out) {
  if (!cfg || !out) {
    set_error("mimo_plan_create: null argument");
    return MIMO_ERR_INVALID;
  }
  MIMO_HIP_CHECK(hipSetDevice(cfg->device));
  auto p = std::make_unique<mimo_plan>();
  p->cfg = *cfg;
  const int rc = p->build();
  if (rc != MIMO_OK) return rc;
  MIMO_HIP_CHECK(hipDeviceSynchronize());
  *out = p.release();
  return MIMO_OK;
}

void mimo_plan_destroy(mimo_plan* plan) { delete plan; }
size_t mimo_plan_workspace_bytes(const mimo_plan* plan) { return plan ? plan->bytes : 0; }
int mimo_plan_num_tensors(const mimo_plan* plan) { return plan ? (int)plan->tensors.size() : 0; }
int64_t mimo_plan_param_floats(const mimo_plan* plan) { return plan ? plan->param_floats : 0; }
int64_t mimo_plan_buffer_floats(const mimo_plan* plan) { return plan ? plan->buffer_floats : 0; }
int mimo_plan_num_double_convs(const mimo_plan* plan) { return plan ? (int)plan->dcs.size() : 0; }
int mimo_plan_double_conv_channels(const mimo_plan* plan, int index) {
  if (!plan || index < 0 || index >= (int)plan->dcs.size()) return -1;
  return plan->dcs[index]->c2.Cout;
}

int mimo_plan_tensor_info(const mimo_plan* plan, int index, char* name, int name_cap, int64_t shape[4], int* ndim,
                          int* kind, int64_t* offset) {
  if (!plan || index < 0 || index >= (int)plan->tensors.size()) {
    set_error("mimo_plan_tensor_info: bad index");
    return MIMO_ERR_INVALID;
  }
  const TensorInfo& t = plan->tensors[index];
  if (name && name_cap > 0) {
    strncpy(name, t.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (shape)
    for (int i = 0; i < 4; ++i) shape[i] = t.shape[i];
  if (ndim) *ndim = t.ndim;
  if (kind) *kind = t.kind;
  if (offset) *offset = t.offset;
  return MIMO_OK;
}

int mimo_plan_bind(mimo_plan* plan, float* params, float* grads, float* bn_buffers) {
  if (!plan || !params || !bn_buffers) {
    set_error("mimo_plan_bind: null argument");
    return MIMO_ERR_INVALID;
  }
  if (plan->params != params || plan->bnbuf != bn_buffers) {
    plan->derived_version = -1;  // other tensors: re-derive
    plan->drop_graphs();         // the captured kernels hold the old parameter pointers
  }
  if (plan->grads != grads) plan->drop_graphs();
  plan->params = params;
  plan->grads = grads;
  plan->bnbuf = bn_buffers;
  return MIMO_OK;
}

int mimo_plan_status(mimo_plan* plan, int32_t* flags, int32_t clear, mimo_stream stream) {
  if (!plan || !flags) {
    set_error("mimo_plan_status: bad argument");
    return MIMO_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  int v = 0;
  MIMO_HIP_CHECK(hipMemcpyAsync(&v, plan->d_status, sizeof(int), hipMemcpyDeviceToHost, st));
  if (clear) MIMO_HIP_CHECK(hipMemsetAsync(plan->d_status, 0, sizeof(int), st));
  MIMO_HIP_CHECK(hipStreamSynchronize(st));
  *flags = v;
  return MIMO_OK;
}

int mimo_plan_dropout_mask(mimo_plan* plan, int site, float* dst, mimo_stream stream) {
  if (!plan || !dst || site < 0 || site >= (int)plan->dcs.size() + 1 + plan->S) {
    set_error("mimo_plan_dropout_mask: bad argument");
    return MIMO_ERR_INVALID;
  }
  const int ndc = (int)plan->dcs.size();
  if (site < ndc) {  // Dropout2d: the multipliers the last forward used (drawn in the engine or staged from the caller)
    const float* src = plan->dcs[site]->mask;
    const size_t n = (size_t)plan->N * plan->dcs[site]->c2.Cout;
    if (!src) {
      set_error("mimo_plan_dropout_mask: site %d had no mask in the last forward", site);
      return MIMO_ERR_STATE;
    }
    MIMO_HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MIMO_OK;
  }
  const int j = site - ndc;
  if (!plan->elem_rng_on[j]) {
    set_error("mimo_plan_dropout_mask: element-wise site %d was not drawn by the engine in the last forward", j);
    return MIMO_ERR_STATE;
  }
  const Act& o = j == 0 ? plan->down4->out : plan->up4[j - 1]->out;
  const ElemRng g = plan->elem_rng(j);
  return elem_dropout_mask_launch(dst, plan->N, o.C, o.Cp, o.H * o.W, g.site, g.seed, g.offset, g.p, (hipStream_t)stream);
}

int mimo_plan_profile(mimo_plan* plan, int enable) {
  if (!plan) {
    set_error("mimo_plan_profile: null plan");
    return MIMO_ERR_INVALID;
  }
  MIMO_TRY(plan->prof_collect());
  plan->prof_on = enable != 0;
  if (enable) {
    for (int k = 0; k < MIMO_PROF_KINDS; ++k) {
      plan->prof_ms[k] = plan->prof_flops[k] = plan->prof_bytes[k] = 0.0;
      plan->prof_launches[k] = 0;
    }
    for (double& v : plan->prof_tier_ms) v = 0.0;
    for (auto& row : plan->prof_kind_tier_ms)
      for (double& v : row) v = 0.0;
  }
  return MIMO_OK;
}

int mimo_plan_profile_read(mimo_plan* plan, int kind, double* total_ms, int64_t* launches, double* flops, double* bytes) {
  if (!plan || kind < 0 || kind >= MIMO_PROF_KINDS) {
    set_error("mimo_plan_profile_read: bad argument");
    return MIMO_ERR_INVALID;
  }
  MIMO_TRY(plan->prof_collect());
  if (total_ms) *total_ms = plan->prof_ms[kind];
  if (launches) *launches = plan->prof_launches[kind];
  if (flops) *flops = plan->prof_flops[kind];
  if (bytes) *bytes = plan->prof_bytes[kind];
  return MIMO_OK;
}

int mimo_plan_profile_read_tier(mimo_plan* plan, int tier, double* forward_ms, double* backward_ms) {
  if (!plan || tier < 0 || tier >= mimo_plan::kProfTiers) {
    set_error("mimo_plan_profile_read_tier: bad argument");
    return MIMO_ERR_INVALID;
  }
  MIMO_TRY(plan->prof_collect());
  if (forward_ms) *forward_ms = plan->prof_tier_ms[2 * tier];
  if (backward_ms) *backward_ms = plan->prof_tier_ms[2 * tier + 1];
  return MIMO_OK;
}

int mimo_plan_profile_read_kind_tier(mimo_plan* plan, int kind, int tier, double* ms) {
  if (!plan || kind < 0 || kind >= MIMO_PROF_KINDS || tier < 0 || tier >= mimo_plan::kProfTiers || !ms) {
    set_error("mimo_plan_profile_read_kind_tier: bad argument");
    return MIMO_ERR_INVALID;
  }
  MIMO_TRY(plan->prof_collect());
  *ms = plan->prof_kind_tier_ms[kind][tier];
  return MIMO_OK;
}

int mimo_forward(mimo_plan* plan, const mimo_forward_args* args, mimo_stream stream) {
  if (!plan) {
    set_error("mimo_forward: null plan");
    return MIMO_ERR_INVALID;
  }
  return plan->forward(args, (hipStream_t)stream);
}

int mimo_loss_forward(mimo_plan* plan, const float* label, const float* mask, const int64_t* perm, float* loss_out,
                      mimo_stream stream) {
  if (!plan) {
    set_error("mimo_loss_forward: null plan");
    return MIMO_ERR_INVALID;
  }
  return plan->loss_forward(label, mask, perm, loss_out, (hipStream_t)stream);
}

int mimo_backward(mimo_plan* plan, const float* dout, const float* dloss, float* dx, mimo_stream stream) {
  if (!plan) {
    set_error("mimo_backward: null plan");
    return MIMO_ERR_INVALID;
  }
  return plan->backward(dout, dloss, dx, 0, mimo_plan::kBwdStages - 1, (hipStream_t)stream);
}

int mimo_backward_stage(mimo_plan* plan, int stage, const float* dout, const float* dloss, float* dx, mimo_stream stream) {
  if (!plan) {
    set_error("mimo_backward_stage: null plan");
    return MIMO_ERR_INVALID;
  }
  return plan->backward(dout, dloss, dx, stage, stage, (hipStream_t)stream);
}

int mimo_backward_stage_async(mimo_plan* plan, int stage, const float* dout, const float* dloss, float* dx, mimo_stream stream,
                              mimo_stream* ready_stream) {
  if (!plan || !ready_stream) {
    set_error("mimo_backward_stage_async: null argument");
    return MIMO_ERR_INVALID;
  }
  hipStream_t ready = nullptr;
  const int rc = plan->backward(dout, dloss, dx, stage, stage, (hipStream_t)stream, &ready);
  *ready_stream = (mimo_stream)ready;
  return rc;
}

int64_t mimo_plan_encoder_param_floats(const mimo_plan* plan) { return plan ? plan->encoder_param_floats : 0; }

int mimo_plan_num_backward_stages(const mimo_plan* plan) { return plan ? mimo_plan::kBwdStages : 0; }

int mimo_plan_backward_stage_range(const mimo_plan* plan, int stage, int64_t* begin, int64_t* end) {
  if (!plan || stage < 0 || stage >= mimo_plan::kBwdStages || !begin || !end) {
    set_error("mimo_plan_backward_stage_range: bad argument");
    return MIMO_ERR_INVALID;
  }
  plan->stage_range(stage, begin, end);
  return MIMO_OK;
}

}  // extern "C"
