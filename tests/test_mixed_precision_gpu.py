"""The 16-bit STORAGE modes (include/mimo_hip.h MIMO_PREC_BF16_MIXED / MIMO_PREC_FP16_MIXED = Lightning's "bf16-mixed" /
"16-mixed", the reference's production precision, scripts/train/train_ndvi.py:71): activations, conv outputs and
their gradients live in HBM as bf16 / fp16, convolution operands are that type, fp32 accumulation, fp32 master
weights / BatchNorm statistics / logits / loss / optimiser, loss scaling through torch's GradScaler protocol.

Pinned to the reference (round 4): tests/golden/amp_*.npz hold a training step of the imported reference under
torch.autocast (bf16 and fp16; make_golden.py::amp_fixture), the oracle under the same context reproduces them bit for bit
(tests/test_oracle_golden.py), and the tolerances of the HIP modes are stated in units of the reference's OWN precision
loss in that mode — d = |autocast reference - fp32 reference|, read from the fixtures.  The two runs round at different
points (the engine rounds less: DESIGN 4), so they are two independent noise realisations around the fp32 result: a HIP
mode must sit about as close to fp32 as the reference's own mode does (<= 1.25 d; observed 0.55-1.11 d) and within 1.5 d
of the autocast reference (two independent vectors of length d are sqrt(2) d apart; observed 0.97-1.38 d; training-mode
BatchNorm on these tiny networks amplifies single rounding-boundary flips, hence the spread).  The older checks remain: (a) the oracle with the engine's
rounding points inserted (`O.conv_operands`), (b) the fp32 goldens / fp32 oracle at mixed-precision tolerance."""
import numpy as np
import pytest
import torch

from oracle import mimo_oracle as O
from tests.helpers import AMP_CASES, amp_reference, cfg_from_meta, grads_rel_l2, load_npz, rel_err, report, state_from
from tests.test_network_gpu import build_model, is_prebn_bias

pytestmark = pytest.mark.gpu
MODES = ["bf16-mixed", "16-mixed"]


def _grad_cosine(model, ref_grads):
    dot = n1 = n2 = 0.0
    for k, p in model.named_parameters():
        k = k[len("model."):]
        if is_prebn_bias(k):
            continue
        a, b = p.grad.detach().cpu().double(), torch.as_tensor(ref_grads[k]).double()
        dot, n1, n2 = dot + float((a * b).sum()), n1 + float((a * a).sum()), n2 + float((b * b).sum())
    return dot / (n1 * n2) ** 0.5, (n1 / n2) ** 0.5


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["cfg1_step.npz", "mini_s2_step.npz"])
def test_storage_modes_on_the_reference_goldens(name, mode):
    fx = load_npz(name)
    cfg = cfg_from_meta(fx["meta"])
    image, label, perms = (torch.from_numpy(fx[f"s0/{k}"]) for k in ("image", "label", "perms"))
    mask = torch.from_numpy(fx["s0/mask"]) if "s0/mask" in fx else None
    S, N, half = cfg.num_subnetworks, image.shape[0], cfg.out_channels // 2
    model = build_model(cfg, state_from(fx, "init/"), loss=str(fx["loss_kind"]), lr=float(fx["lr"]), wd=float(fx["wd"]),
                        T=float(fx["temperature"]), precision=mode)
    # (a) eval-mode forward: grad-enabled path (z stored, separate BatchNorm pass) vs the oracle with the same roundings
    x = torch.stack([image[perms[s]] for s in range(S)], dim=1)
    model.eval()
    p1, p2 = model(x.cuda())
    with torch.no_grad():
        q1, q2 = model(x.cuda())  # inference path: BatchNorm folded into the conv epilogue (one rounding fewer)
        with O.conv_operands(mode):
            o16 = O.mimo_unet_forward(cfg, state_from(fx, "init/"), x, training=False)
        o32 = O.mimo_unet_forward(cfg, state_from(fx, "init/"), x, training=False)
    hip = torch.cat([p1, p2], dim=2).detach().cpu()
    e16, e32 = rel_err(hip, o16), rel_err(hip, o32)
    e_inf = rel_err(torch.cat([q1, q2], dim=2).cpu(), o32)
    # (b) one training step against the fp32 reference golden; fp16: under a fixed loss scale of 1024
    model.train()
    scale = 1024.0 if mode == "16-mixed" else 1.0
    out = model.training_step_with_perms(image.cuda(), label.cuda(), None if mask is None else mask.cuda(), perms.cuda())
    (out["loss"] * scale).backward()
    for p in model.parameters():
        p.grad.div_(scale)
    preds = out["preds"].view(N, S, half, *image.shape[-2:]).cpu()
    t32 = rel_err(preds, torch.from_numpy(fx["s0/out"])[:, :, :half])
    ref = {k[len("s0/grad/"):]: v for k, v in fx.items() if k.startswith("s0/grad/")}
    cos, ratio = _grad_cosine(model, ref)
    report(f"{mode} {name}: eval fwd vs emulating oracle {e16:.2e}, vs fp32 oracle {e32:.2e} (inference path {e_inf:.2e}); "
           f"train out vs fp32 golden {t32:.2e}; gradient cosine {cos:.4f}, |g|/|g_ref| {ratio:.4f}")
    lim = 2e-2 if mode == "bf16-mixed" else 3e-3   # 8 vs 11 mantissa bits
    assert e16 < lim and e32 < 2 * lim and e_inf < 2 * lim
    assert t32 < 3e-1 and cos > (0.8 if mode == "bf16-mixed" else 0.95) and 0.85 < ratio < 1.15
    np.testing.assert_allclose(out["loss"].item(), fx["s0/total"], rtol=5e-2, atol=5e-3)
    assert all(torch.isfinite(p.grad).all() for p in model.parameters())


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("amp_name,src", AMP_CASES)
def test_storage_modes_against_the_reference_under_autocast(amp_name, src, mode):
    """The HIP 16-bit storage modes against the REFERENCE run under torch.autocast (tests/golden/amp_*.npz: Lightning's
    precision="16-mixed", scripts/train/train_ndvi.py:71, and its bf16 twin), same parameters, inputs, shuffles and loss
    scale.  d_* = the reference's own distance between its autocast and its fp32 result on that quantity (fixtures)."""
    r = amp_reference(amp_name, src, mode)
    cfg, fx = r["cfg"], r["fx"]
    S, N, half = cfg.num_subnetworks, r["image"].shape[0], cfg.out_channels // 2
    model = build_model(cfg, state_from(fx, "init/"), T=float(fx["temperature"]), precision=mode)
    model.train()
    out = model.training_step_with_perms(r["image"].cuda(), r["label"].cuda(), None if r["mask"] is None else r["mask"].cuda(),
                                         r["perms"].cuda())
    (out["loss"] * r["scale"]).backward()
    hip = {k[len("model."):]: p.grad.detach().cpu() / r["scale"] for k, p in model.named_parameters()}
    preds = out["preds"].view(N, S, half, *r["image"].shape[-2:]).cpu()
    e_out16, e_out32 = rel_err(preds, r["out16"][:, :, :half]), rel_err(preds, r["out32"][:, :, :half])
    d_out = rel_err(r["out16"][:, :, :half], r["out32"][:, :, :half])
    e_g16, e_g32 = grads_rel_l2(hip, r["grads16"]), grads_rel_l2(hip, r["grads32"])
    e_loss = abs(out["loss"].item() - r["total16"]) / abs(r["total16"])
    # eval mode with the initial running statistics (no batch statistics to amplify a rounding)
    model2 = build_model(cfg, state_from(fx, "init/"), precision=mode)
    model2.eval()
    x = torch.stack([r["image"][r["perms"][s]] for s in range(S)], dim=1)
    with torch.no_grad():
        q1, q2 = model2(x.cuda())
        o32 = O.mimo_unet_forward(cfg, state_from(fx, "init/"), x, training=False)
    q = torch.cat([q1, q2], dim=2).cpu()
    d_eval, e_ev16, e_ev32 = rel_err(r["eval16"], o32), rel_err(q, r["eval16"]), rel_err(q, o32)
    report(f"{mode} {amp_name} vs the reference under autocast: train out {e_out16:.2e} (d {d_out:.2e}; vs fp32 ref {e_out32:.2e}); "
           f"gradient rel-L2 {e_g16:.2e} (d {r['d_grads']:.2e}; vs fp32 ref {e_g32:.2e}); loss {e_loss:.2e}; "
           f"eval out {e_ev16:.2e} (d {d_eval:.2e}; vs fp32 oracle {e_ev32:.2e})")
    assert e_out32 <= 1.25 * d_out and e_g32 <= 1.25 * r["d_grads"]    # about as close to fp32 as the reference's own mode
    assert e_out16 <= 1.5 * d_out and e_g16 <= 1.5 * r["d_grads"]      # two independent roundings: ~sqrt(2) d apart
    assert e_loss < 5e-3
    assert e_ev32 <= 0.5 * d_eval and e_ev16 <= 1.25 * d_eval


@pytest.mark.parametrize("mode,H", [("bf16-mixed", 256), ("16-mixed", 128)])
def test_storage_modes_against_the_autocast_oracle_at_cfg3_widths(mode, H):
    """cfg3's widths (2 -> 1 ch, S = 2, fbc = 30: 120 ... 960-channel core) against the oracle run under the reference's
    autocast context (`O.reference_autocast`, pinned bit for bit to tests/golden/amp_*.npz) on this box's host — the
    reference-precision counterpart of the per-tensor emulation test below.  256 x 256 for bf16; 128 x 128 for fp16,
    whose CPU convolutions are ~40 x slower.  Bounds in units of d = autocast oracle vs fp32 oracle, as above."""
    cfg = O.NetConfig(2, 2, 2, 30)
    N, S = 2, 2
    g = torch.Generator().manual_seed(33)
    st = O.init_state(cfg, 33)
    image, label = torch.rand(N, 2, H, H, generator=g), torch.rand(N, 1, H, H, generator=g)
    perms = O.draw_perms(N, S, generator=g)
    scale = 1024.0 if mode == "16-mixed" else 1.0
    lb_w = torch.tensor([0.8, 1.2])
    model = build_model(cfg, st, precision=mode)
    model.train()
    model.loss_buffer.get_weights = lambda: lb_w
    out = model.training_step_with_perms(image.cuda(), label.cuda(), None, perms.cuda())
    (out["loss"] * scale).backward()
    hip = {k[len("model."):]: p.grad.detach().cpu() / scale for k, p in model.named_parameters()}

    def oracle(autocast):
        ts = O.TrainState(cfg=cfg, st={k: v.clone() for k, v in st.items()}, loss_buffer=O.LossBuffer(S, 0.3, 10))
        ts.loss_buffer.get_weights = lambda: lb_w
        return O.train_step(ts, image, label, None, perms, apply_optimizer=False, loss_scale=scale, autocast=autocast)

    a16, f32 = oracle(mode), oracle(None)
    preds = out["preds"].view(N, S, 1, H, H).cpu()
    d_out = rel_err(a16["out"][:, :, :1].float(), f32["out"][:, :, :1])
    e_out16, e_out32 = rel_err(preds, a16["out"][:, :, :1].float()), rel_err(preds, f32["out"][:, :, :1])
    d_g = grads_rel_l2(a16["grads"], f32["grads"])
    e_g16, e_g32 = grads_rel_l2(hip, a16["grads"]), grads_rel_l2(hip, f32["grads"])
    report(f"{mode} cfg3 widths {H}x{H} N=2 vs the autocast oracle: train out {e_out16:.2e} (d {d_out:.2e}; vs fp32 {e_out32:.2e}); "
           f"gradient rel-L2 {e_g16:.2e} (d {d_g:.2e}; vs fp32 {e_g32:.2e})")
    assert e_out32 <= 1.25 * d_out and e_g32 <= 1.25 * d_g
    assert e_out16 <= 1.5 * d_out and e_g16 <= 1.5 * d_g


def test_cfg4_geometry_bf16_mixed_vs_oracle_and_memory():
    """BASELINE config 4 (S=4, fbc=30, bf16) in the bf16 STORAGE mode at 256x256: parity against the oracle, and the
    activation workspace really is 16-bit (about half of the fp32-storage plan)."""
    cfg = O.NetConfig(2, 2, 4, 30)
    N, H, W, S = 2, 256, 256, 4
    g = torch.Generator().manual_seed(24)
    st = O.init_state(cfg, 24)
    image = torch.rand(N, 2, H, W, generator=g)
    label = torch.rand(N, 1, H, W, generator=g)
    perms = O.draw_perms(N, S, generator=g)
    model = build_model(cfg, st, precision="bf16-mixed")
    x = torch.stack([image[perms[s]] for s in range(S)], dim=1)
    model.eval()
    p1, p2 = model(x.cuda())
    with torch.no_grad():
        with O.conv_operands("bf16-mixed"):
            o16 = O.mimo_unet_forward(cfg, st, x, training=False)
        o32 = O.mimo_unet_forward(cfg, st, x, training=False)
    hip = torch.cat([p1, p2], dim=2).detach().cpu()
    e16, e32 = rel_err(hip, o16), rel_err(hip, o32)
    del p1, p2  # drop the eval-mode autograd graph: its plan is free again (else the training step gets a second plan)
    model.train()
    lb_w = torch.tensor([0.7, 0.9, 1.1, 1.3])
    model.loss_buffer.get_weights = lambda: lb_w
    out = model.training_step_with_perms(image.cuda(), label.cuda(), None, perms.cuda())
    out["loss"].backward()
    ts = O.TrainState(cfg=cfg, st={k: v.clone() for k, v in st.items()}, loss_buffer=O.LossBuffer(S, 0.3, 10))
    ts.loss_buffer.get_weights = lambda: lb_w
    ref = O.train_step(ts, image, label, None, perms, apply_optimizer=False)
    preds = out["preds"].view(N, S, 1, H, W).cpu()
    t32 = rel_err(preds, ref["out"][:, :, :1])
    e_loss = abs(out["loss"].item() - float(ref["total"])) / abs(float(ref["total"]))
    cos, ratio = _grad_cosine(model, ref["grads"])
    ws16 = sum(p.workspace_bytes for p in model.model._plans.values())
    m32 = build_model(cfg, st, precision="bf16")
    m32.train()
    m32.training_step_with_perms(image.cuda(), label.cuda(), None, perms.cuda())
    ws32 = sum(p.workspace_bytes for p in m32.model._plans.values())
    report(f"cfg4 256x256 N=2 [bf16-mixed]: eval fwd vs emulating oracle {e16:.2e}, vs fp32 oracle {e32:.2e}; train out vs "
           f"fp32 oracle {t32:.2e}, loss {e_loss:.2e}, gradient cosine {cos:.5f}, |g|/|g_ref| {ratio:.4f}; plan workspace "
           f"{ws16 / 2**30:.2f} GiB vs {ws32 / 2**30:.2f} GiB with fp32 storage")
    assert e16 < 2e-2 and e32 < 4e-2 and t32 < 2e-1 and e_loss < 2e-2 and cos > 0.85 and 0.9 < ratio < 1.1
    assert ws16 < 0.75 * ws32  # activations halve; packed weights, weight-gradient slabs and staging buffers do not


@pytest.mark.parametrize("mode", MODES)
def test_storage_mode_gradients_anchor_on_the_emulating_oracle_at_cfg3_geometry(mode):
    """cfg3's geometry (2 -> 1 ch, S = 2, fbc = 30, 256 x 256; N = 2) in both storage modes — `16-mixed` is the
    reference's production precision (scripts/train/train_ndvi.py:71).  The backward is anchored PER TENSOR on the
    oracle that emulates the engine's rounding points in the forward AND the backward (`conv_operands(mode,
    grad_storage=True)`: operands, stored activations / conv outputs, their gradients and the padded-domain data
    gradient), run under the same fixed loss scale — not on a cosine against the fp32 run.  What remains between the
    two is summation order and the rounding-boundary / ReLU-mask flips it causes (training-mode BatchNorm amplifies
    them), so the bound is stated relative to the distance of the emulation itself from the fp32 oracle: every tensor
    and the whole gradient must sit much closer to the emulation than the emulation sits to fp32."""
    cfg = O.NetConfig(2, 2, 2, 30)
    N, H, W, S = 2, 256, 256, 2
    g = torch.Generator().manual_seed(31)
    st = O.init_state(cfg, 31)
    image = torch.rand(N, 2, H, W, generator=g)
    label = torch.rand(N, 1, H, W, generator=g)
    perms = O.draw_perms(N, S, generator=g)
    scale = 1024.0 if mode == "16-mixed" else 1.0
    lb_w = torch.tensor([0.8, 1.2])
    model = build_model(cfg, st, precision=mode)
    model.train()
    model.loss_buffer.get_weights = lambda: lb_w
    out = model.training_step_with_perms(image.cuda(), label.cuda(), None, perms.cuda())
    (out["loss"] * scale).backward()
    hip = {k[len("model."):]: p.grad.detach().cpu().double() / scale for k, p in model.named_parameters()}

    def oracle(kind, **kw):
        ts = O.TrainState(cfg=cfg, st={k: v.clone() for k, v in st.items()}, loss_buffer=O.LossBuffer(S, 0.3, 10))
        ts.loss_buffer.get_weights = lambda: lb_w
        with O.conv_operands(kind, **kw):
            return O.train_step(ts, image, label, None, perms, apply_optimizer=False, loss_scale=scale)

    emu, f32 = oracle(mode, grad_storage=True), oracle("fp32")
    preds = out["preds"].view(N, S, 1, H, W).cpu()
    e_out_emu, e_out_32 = rel_err(preds, emu["out"][:, :, :1]), rel_err(preds, f32["out"][:, :, :1])
    worst, num_e, num_f, den = ("", 0.0, 0.0), 0.0, 0.0, 0.0
    for k, ge in emu["grads"].items():
        if is_prebn_bias(k):
            continue
        ge, gf = ge.double(), f32["grads"][k].double()
        e_emu = float((hip[k] - ge).norm() / ge.norm())
        e_f32 = float((ge - gf).norm() / gf.norm())  # how far the emulation itself sits from fp32 on this tensor
        if e_emu > worst[1]:
            worst = (k, e_emu, e_f32)
        num_e += float(((hip[k] - ge) ** 2).sum())
        num_f += float(((ge - gf) ** 2).sum())
        den += float((ge ** 2).sum())
        assert e_emu <= 2e-2 + 1.0 * e_f32, (k, e_emu, e_f32)
    rel_emu, rel_f32 = (num_e / den) ** 0.5, (num_f / den) ** 0.5
    report(f"{mode} cfg3 256x256 N=2: train out vs emulating oracle {e_out_emu:.2e} (vs fp32 oracle {e_out_32:.2e}); gradient "
           f"rel-L2 vs emulating oracle {rel_emu:.2e} (emulation vs fp32 {rel_f32:.2e}); worst tensor {worst[0]} "
           f"{worst[1]:.2e} (emulation vs fp32 there {worst[2]:.2e})")
    assert e_out_emu < 1e-1 and rel_emu < 0.75 * rel_f32 + 1e-2  # observed 0.57 (bf16) / 0.69 (fp16) of it


def _amp_model(mode="16-mixed"):
    fx = load_npz("mini_s2_step.npz")
    cfg = cfg_from_meta(fx["meta"])
    model = build_model(cfg, state_from(fx, "init/"), precision=mode, lr=1e-3)
    model.train()
    batch = [torch.from_numpy(fx[f"s0/{k}"]).cuda() for k in ("image", "label", "perms")]
    return model, batch


def test_fp16_mixed_training_under_torch_grad_scaler():
    """precision="16-mixed" as Lightning runs it: torch's GradScaler scales the loss, FlatAdam (a fused optimiser in
    GradScaler's protocol: `_step_supports_amp_scaling`) divides the gradients by the scale, skips the update on inf /
    nan and counts its steps on the device.  Three scaled steps track the fp32-class (split16) trajectory; a scale
    that overflows fp16 leaves parameters and step count untouched and halves the scale."""
    ref_model, (image, label, perms) = _amp_model("split16")
    ref_opt = ref_model.configure_optimizers()["optimizer"]
    model, _ = _amp_model("16-mixed")
    opt = model.configure_optimizers()["optimizer"]
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 12, growth_interval=2)
    for it in range(3):
        for m, o, sc in ((ref_model, ref_opt, None), (model, opt, scaler)):
            o.zero_grad()
            loss = m.training_step_with_perms(image, label, None, perms)["loss"]
            if sc is None:
                loss.backward()
                o.step()
            else:
                sc.scale(loss).backward()
                sc.step(o)
                sc.update()
    assert opt.step_count == 3 and scaler.get_scale() == 2.0 ** 13  # grew once after 2 clean steps
    a, b = model.model.flat_parameters(), ref_model.model.flat_parameters()
    d = float((a - b).abs().max())
    report(f"16-mixed + GradScaler, 3 Adam steps on mini_s2: max |param - split16 param| {d:.2e} (lr 1e-3: 3 steps move a parameter by <= 3e-3)")
    # Adam turns a pure-noise gradient (a conv bias in front of BatchNorm) into +-lr steps: two arithmetics can walk such
    # a parameter in opposite directions on every step, so the MAXIMUM sits at the step budget 2 x 3 x lr by construction
    # (observed 5.96e-3).  The bound on it therefore carries a real margin (10 %), and the test is carried by the bulk:
    # the rms difference and the share of parameters that moved apart by more than one step budget
    frac_far = float(((a - b).abs() > 3e-3).float().mean())
    report(f"16-mixed + GradScaler: rms difference {float((a - b).pow(2).mean().sqrt()):.2e}, share beyond 3e-3: {frac_far:.4f}")
    assert d <= 2.2 * 3 * 1e-3
    assert float((a - b).pow(2).mean().sqrt()) < 0.5 * 3e-3 and frac_far < 0.10
    # overflow: a scale of 2^40 pushes the scaled gradients out of fp16's range
    before, steps_before = a.clone(), opt.step_count
    big = torch.amp.GradScaler("cuda", init_scale=2.0 ** 40)
    opt.zero_grad()
    big.scale(model.training_step_with_perms(image, label, None, perms)["loss"]).backward()
    big.step(opt)
    big.update()
    assert torch.equal(model.model.flat_parameters(), before) and opt.step_count == steps_before
    assert big.get_scale() == 2.0 ** 39
    assert not hasattr(opt, "grad_scale") and not hasattr(opt, "found_inf")  # the scaler removed its attributes
    # and the optimiser still steps without a scaler afterwards (device counter keeps counting)
    opt.zero_grad()
    model.training_step_with_perms(image, label, None, perms)["loss"].backward()
    opt.step()
    assert opt.step_count == steps_before + 1 and not torch.equal(model.model.flat_parameters(), before)


@pytest.mark.parametrize("mode", MODES)
def test_storage_modes_dropout_odd_size_and_input_gradient(mode):
    """Odd image size (floor pooling, zero F.pad in the up-sampling), Dropout2d masks, loss mask and the input gradient
    through the 16-bit storage path: finite, and close to the fp32-class result of the same masks."""
    cfg = O.NetConfig(3, 2, 2, 6)
    st = O.init_state(cfg, 5)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 2, 3, 50, 70, generator=g).cuda()
    y = torch.rand(2, 2, 1, 50, 70, generator=g).cuda()
    res = {}
    for prec in ("split16", mode):
        model = build_model(cfg, st, precision=prec, dropout=(0.2, 0.2, 0.2))
        model.train()
        torch.cuda.manual_seed(3)  # same Dropout2d draws
        xg = x.clone().requires_grad_(True)
        p1, p2 = model(xg)
        loss = model.loss_fn.forward(p1, p2, y, reduce_mean=False).mean(dim=(0, 2, 3, 4)).mean()
        (loss * 256.0).backward()
        res[prec] = (p1.detach(), xg.grad / 256.0, model.model.flat_gradients() / 256.0)
    e_out = rel_err(res[mode][0].cpu(), res["split16"][0].cpu())
    gx = torch.nn.functional.cosine_similarity(res[mode][1].flatten(), res["split16"][1].flatten(), dim=0).item()
    gw = torch.nn.functional.cosine_similarity(res[mode][2], res["split16"][2], dim=0).item()
    report(f"{mode} 50x70 dropout: out vs split16 {e_out:.2e}, cosine dx {gx:.4f}, cosine dW {gw:.4f}")
    assert all(torch.isfinite(t).all() for t in res[mode])
    assert e_out < 2e-1 and gx > 0.8 and gw > 0.8
