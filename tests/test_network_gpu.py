"""Whole-path parity on the MI355X: the MimoUnetModel surface (HIP engine through the C ABI)
against (a) golden vectors generated from the imported reference and (b) the CPU oracle run
live on the same seeded inputs.  Tolerance: the north_star's 1e-3 relative (fp32), measured
as max|a-b| / max|b|; observed errors are printed."""
import numpy as np
import pytest
import torch

from oracle import mimo_oracle as O
from tests.helpers import (adam_flip_bound, adam_flip_statistic, cfg_from_meta, golden_grad_tol, load_npz, rel_err, report,
                           state_from, wgrad_two_mfma)

pytestmark = pytest.mark.gpu
TOL = 1e-3


PRECISIONS = ["split16", "fp32"]


def build_model(cfg: O.NetConfig, state, *, loss="laplace_nll", lr=1e-3, wd=0.0, T=0.3, dropout=(0.0, 0.0, 0.0),
                precision="split16", center_final=(0.0, 0.0)):
    from mimo.models.mimo_unet import MimoUnetModel
    m = MimoUnetModel(in_channels=cfg.in_channels, out_channels=cfg.out_channels, num_subnetworks=cfg.num_subnetworks,
                      filter_base_count=cfg.filter_base_count, center_dropout_rate=center_final[0],
                      final_dropout_rate=center_final[1],
                      encoder_dropout_rate=dropout[0], core_dropout_rate=dropout[1], decoder_dropout_rate=dropout[2],
                      loss=loss, weight_decay=wd, learning_rate=lr, seed=0, loss_buffer_size=10, loss_buffer_temperature=T)
    m.load_state_dict({"model." + k: v for k, v in state.items()})
    m.model.set_precision(precision)
    return m.cuda()


def is_prebn_bias(k):
    return k.endswith((".0.bias", ".3.bias")) and "double_conv" in k


def check_grads(named_grads, ref_grads, tol=TOL):
    """Per-tensor max error relative to the tensor's scale.  The north_star's tolerance is 1e-3; the checks against the
    reference GOLDENS pass `golden_grad_tol(precision)` — 3e-4 for the fp32 mode, 5e-4 with the three-MFMA weight gradient,
    8e-4 with the two-MFMA weight gradient (observed <= 1.7e-4 / 3.6e-4 / 6.2e-4, tests/helpers.py): a precision
    trade shows as a red test before it reaches the headline tolerance."""
    worst = ("", 0.0)
    for k, ref in ref_grads.items():
        g = named_grads[k].double()
        r = torch.as_tensor(ref).double()
        scale = ref_grads[k[:-4] + "weight"] if is_prebn_bias(k) else ref  # zero-gradient biases: noise, see oracle test
        e = float((g - r).abs().max()) / max(float(torch.as_tensor(scale).abs().max()), 1e-30)
        if e > worst[1]:
            worst = (k, e)
        assert e < tol, (k, e)
    return worst


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", ["cfg1_step.npz", "mini_s2_step.npz", "mini_gauss_step.npz"])
def test_train_steps_match_reference_golden(name, precision):
    fx = load_npz(name)
    cfg = cfg_from_meta(fx["meta"])
    steps, lr, wd = int(fx["meta"][8]), float(fx["lr"]), float(fx["wd"])
    model = build_model(cfg, state_from(fx, "init/"), loss=str(fx["loss_kind"]), lr=lr, wd=wd, T=float(fx["temperature"]),
                        precision=precision)
    model.train()
    opt = model.configure_optimizers()["optimizer"]
    for it in range(steps):
        image = torch.from_numpy(fx[f"s{it}/image"]).cuda()
        label = torch.from_numpy(fx[f"s{it}/label"]).cuda()
        mask = torch.from_numpy(fx[f"s{it}/mask"]).cuda() if f"s{it}/mask" in fx else None
        perms = torch.from_numpy(fx[f"s{it}/perms"]).cuda()
        opt.zero_grad()
        out_dict = model.training_step_with_perms(image, label, mask, perms)
        out_dict["loss"].backward()
        np.testing.assert_allclose(out_dict["loss"].item(), fx[f"s{it}/total"], rtol=TOL)
        np.testing.assert_allclose(model.logged["train_loss"].item(), fx[f"s{it}/loss"].mean(), rtol=TOL)
        for s in range(cfg.num_subnetworks):
            np.testing.assert_allclose(model.logged[f"train_loss_{s}"].item(), fx[f"s{it}/loss"][s], rtol=TOL, atol=1e-6)
            np.testing.assert_allclose(model.logged[f"train_weight_{s}"].item(), fx[f"s{it}/weights"][s], rtol=TOL)
        if it == 0:
            half = cfg.out_channels // 2
            S, N = cfg.num_subnetworks, image.shape[0]
            ref_out = torch.from_numpy(fx["s0/out"])
            preds = out_dict["preds"].view(N, S, half, *image.shape[-2:]).cpu()
            e_out = rel_err(preds, ref_out[:, :, :half])
            grads = {k[len("model."):]: p.grad.detach().cpu() for k, p in model.named_parameters()}
            worst = check_grads(grads, {k[len("s0/grad/"):]: v for k, v in fx.items() if k.startswith("s0/grad/")},
                                tol=golden_grad_tol(precision))
            report(f"{name} [{precision}]: out err {e_out:.2e}, worst grad err {worst[1]:.2e} at {worst[0]}")
            assert e_out < TOL
            sd = model.state_dict()
            for k, v in fx.items():
                if k.startswith("s0/after/") and "running" in k:
                    assert rel_err(sd["model." + k[len("s0/after/"):]].cpu(), v) < 1e-4, k
                if k.startswith("s0/after/") and "num_batches" in k:
                    assert int(sd["model." + k[len("s0/after/"):]]) == int(v)
        opt.step()
    sd = model.state_dict()
    budget = steps * lr
    flip_worst = ("", 0.0)
    for k, v in fx.items():
        if not k.startswith("final/") or k == "final/loss_buffer":
            continue
        name_ = k[len("final/"):]
        ours = sd["model." + name_].cpu().numpy()
        if name_.endswith("num_batches_tracked"):
            assert int(ours) == int(v)
        elif is_prebn_bias(name_) or name_.endswith("running_mean"):
            assert np.abs(ours - v).max() <= 2.02 * budget + 1e-5, name_
        elif name_.endswith("running_var"):
            # after `steps` optimiser steps: with the two-MFMA weight gradient (split16 since round 5) the parameters of the
            # later steps differ by Adam's sign flips (below), and the deepest layers normalise over a dozen samples per
            # channel (mini_s2: 3 images of 2 x 2 pixels) — observed 2.3e-3 there; the first step's buffers are held to 1e-4 above
            # (ARITHMETIC-SPECIFIC bound: 5e-3 under the two-MFMA weight gradient, 1e-3 otherwise)
            assert rel_err(ours, v) < (5e-3 if wgrad_two_mfma(precision) else TOL), name_
        else:  # Adam turns rounding-level gradient differences into sign-level update differences
            d = np.abs(ours - v)
            assert d.max() <= 2.02 * budget + 1e-5 * np.abs(v).max(), (name_, d.max())  # two runs, opposite signs
            # rms: how many elements flipped — bounded by what the reference's own fp32 rounding does to it (helpers.ADAM_FLIP_RMS)
            fs = adam_flip_statistic(ours, v, budget)
            if d.size >= 256 and fs > flip_worst[1]:
                flip_worst = (name_, fs)
            assert fs <= (adam_flip_bound(name, precision) if d.size >= 256 else 1.0), (name_, fs)
    report(f"{name} [{precision}]: Adam sign-flip statistic after {steps} steps: worst {flip_worst[1]:.3f} at {flip_worst[0]} "
           f"(bound {adam_flip_bound(name, precision)}; the reference's own fp32 rounding: see tests/helpers.py)")
    np.testing.assert_allclose(model.loss_buffer.buffer.cpu().numpy(), fx["final/loss_buffer"], rtol=TOL, atol=1e-6)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_input_gradient_and_generic_backward_cfg1(precision):
    """dL/d(input image) through the plain forward + torch-side loss (the FGSM path,
    scripts/test/test_nyuv2_depth.py:41-55)."""
    fx = load_npz("cfg1_step.npz")
    cfg = cfg_from_meta(fx["meta"])
    model = build_model(cfg, state_from(fx, "init/"), precision=precision)
    model.train()
    x = torch.from_numpy(fx["s0/image"])[torch.from_numpy(fx["s0/perms"])[0]][:, None].cuda().requires_grad_(True)
    y = torch.from_numpy(fx["s0/label"])[torch.from_numpy(fx["s0/perms"])[0]][:, None].cuda()
    p1, p2 = model(x)
    loss = model.loss_fn.forward(p1, p2, y, reduce_mean=False).mean(dim=(0, 2, 3, 4)).mean()
    loss.backward()
    e = rel_err(x.grad.cpu(), fx["s0/dx"])
    report(f"[{precision}] dx err {e:.2e}")
    assert e < TOL
    grads = {k[len("model."):]: p.grad.detach().cpu() for k, p in model.named_parameters()}
    check_grads(grads, {k[len("s0/grad/"):]: v for k, v in fx.items() if k.startswith("s0/grad/")}, tol=golden_grad_tol(precision))


@pytest.mark.parametrize("precision", PRECISIONS)
def test_elementwise_center_final_dropout_golden(precision):
    """nn.Dropout after down4 and in front of each head (model.py:213, :277-281), recorded masks:
    logits, loss, input gradient and every parameter gradient against the reference."""
    fx = load_npz("elem_dropout.npz")
    Ci, Co, S, f, N, H, W = (int(v) for v in fx["meta"])
    cfg = O.NetConfig(Ci, Co, S, f)
    model = build_model(cfg, state_from(fx, "init/"), precision=precision, center_final=(float(fx["pc"]), float(fx["pf"])))
    model.train()
    model.model.elem_mask_override = {"center": torch.from_numpy(fx["mask/center"]),
                                      **{f"final{s}": torch.from_numpy(fx[f"mask/final{s}"]) for s in range(S)}}
    x = torch.from_numpy(fx["x"]).cuda().requires_grad_(True)
    y = torch.from_numpy(fx["y"]).cuda()
    p1, p2 = model(x)
    e_out = rel_err(torch.cat([p1, p2], dim=2).detach().cpu(), fx["out"])
    loss = model.loss_fn.forward(p1, p2, y, reduce_mean=False).mean(dim=(0, 2, 3, 4))
    np.testing.assert_allclose(loss.detach().cpu().numpy(), fx["loss"], rtol=TOL)
    loss.mean().backward()
    e_dx = rel_err(x.grad.cpu(), fx["dx"])
    grads = {k[len("model."):]: p.grad.detach().cpu() for k, p in model.named_parameters()}
    worst = check_grads(grads, {k[len("grad/"):]: v for k, v in fx.items() if k.startswith("grad/")}, tol=golden_grad_tol(precision))
    report(f"elem dropout [{precision}]: out {e_out:.2e} dx {e_dx:.2e} worst grad {worst[1]:.2e} at {worst[0]}")
    assert e_out < TOL and e_dx < TOL
    # without an override the module draws its own Bernoulli masks: a different, but finite, result
    model.model.elem_mask_override = None
    with torch.no_grad():
        q1, _ = model(x.detach())
    assert torch.isfinite(q1).all() and not torch.equal(q1, p1.detach())
    model.eval()  # nn.Dropout inactive in eval mode
    with torch.no_grad():
        a1, _ = model(x.detach())
        b1, _ = model(x.detach())
    assert torch.equal(a1, b1)


@pytest.mark.parametrize("tag", ["50x70", "100x100", "128x160"])
def test_odd_sizes_forward(tag):
    fx = load_npz("odd_sizes.npz")
    cfg = O.NetConfig(3, 2, 2, 4)
    model = build_model(cfg, state_from(fx, "init/"))
    x = torch.from_numpy(fx[tag + "/x"]).cuda()
    model.train()
    with torch.no_grad():
        p1, p2 = model(x)
    out_train = torch.cat([p1, p2], dim=2).cpu()
    model.eval()
    with torch.no_grad():
        p1, p2 = model(x)
    out_eval = torch.cat([p1, p2], dim=2).cpu()
    e1, e2 = rel_err(out_train, fx[tag + "/out_train"]), rel_err(out_eval, fx[tag + "/out_eval"])
    report(f"odd {tag}: train {e1:.2e} eval {e2:.2e}")
    assert e1 < TOL and e2 < TOL
    sd = model.state_dict()
    for k, v in fx.items():
        if k.startswith(tag + "/after/"):
            assert rel_err(sd["model." + k[len(tag + "/after/"):]].cpu(), v) < 1e-4, k


def test_mc_dropout_ensemble_golden():
    from mimo.models.ensemble import EnsembleModule
    fx = load_npz("mc_dropout.npz")
    Ci, Co, S, f, N, H, W, passes = (int(v) for v in fx["meta"])
    p = float(fx["p"])
    cfg = O.NetConfig(Ci, Co, S, f)
    model = build_model(cfg, state_from(fx, "state/"), dropout=(p, p, p))
    ndc = int(fx["pass0/nmask"])
    # sample (pass m, image i) sits at batch row m*N + i
    model.model.mask_override = {j: torch.cat([torch.from_numpy(fx[f"pass{m}/mask{j}"]) for m in range(passes)], 0)
                                 for j in range(ndc)}
    ens = EnsembleModule([], monte_carlo_steps=passes, models=[model], return_raw_predictions=True)
    x = torch.from_numpy(fx["x"]).cuda()
    p1, p2 = ens(x)
    assert rel_err(p1, fx["p1"]) < TOL and rel_err(p2, fx["p2"]) < TOL
    ens.return_raw_predictions = False
    mean, al, ep = ens(x)
    e = (rel_err(mean, fx["mean"]), rel_err(al, fx["alea"]), rel_err(ep, fx["epi"]))
    report("mc-dropout errs", e)
    assert max(e) < TOL
    assert ens.num_subnetworks == S


def _oracle_run(cfg, st, image, label, mask, perms, lb_w, loss, dtype):
    cast = lambda t: None if t is None else t.to(dtype)
    ts = O.TrainState(cfg=cfg, st={k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in st.items()},
                      loss_kind=loss, loss_buffer=O.LossBuffer(cfg.num_subnetworks, 0.3, 10))
    ts.loss_buffer.get_weights = lambda: lb_w.to(dtype)
    return ts, O.train_step(ts, cast(image), cast(label), cast(mask), perms, apply_optimizer=False)


def _oracle_vs_hip(cfg, N, H, W, seed, loss="laplace_nll", with_mask=False, precision="split16", small_net=False,
                   repetitions=1, irp=0.0):
    """One training step (forward, loss, backward) of the HIP path against the CPU oracle.

    Outputs, losses and BatchNorm buffers: 1e-3 relative against the fp32 oracle.

    Gradients: ReLU / max-pool derivatives are discontinuous, so a single mask flip caused by
    1e-7 forward rounding moves a weight gradient by ~1/sqrt(#pixels) of its scale; the
    oracle's own fp32 run differs from its fp64 run by 3e-3..3e-2 on these shapes (measured,
    tests/tools/diag_grad_noise.py).  The gradient check is therefore anchored on the fp64 oracle:
    the HIP error must stay within the fp32 noise floor measured in the same test
    (rms error <= 1e-3 + 5x the fp32 oracle's rms error, per tensor), and the whole gradient
    must agree in direction and size (cosine > 0.9999, global rel-L2 < 2e-2).  small_net: networks with a
    handful of channels, where one flipped mask moves every upstream tensor by a few 1e-3 (seen in both
    precision modes and, smaller, between the fp32 and fp64 oracles): per-tensor floor 5e-3, cosine > 0.9995."""
    g = torch.Generator().manual_seed(seed)
    st = O.init_state(cfg, seed)
    for k in st:  # non-trivial BN affine parameters
        if (".double_conv.1." in k or ".double_conv.4." in k) and k.endswith("weight"):
            st[k] = 0.5 + torch.rand(st[k].shape, generator=g)
        if (".double_conv.1." in k or ".double_conv.4." in k) and k.endswith(".bias"):
            st[k] = 0.2 * torch.randn(st[k].shape, generator=g)
    image = torch.rand(N, cfg.in_channels, H, W, generator=g)
    label = torch.rand(N, cfg.out_channels // 2, H, W, generator=g)
    mask = (torch.rand(N, 1, H, W, generator=g) > 0.3).float() if with_mask else None
    perms = O.draw_perms(N, cfg.num_subnetworks, irp, repetitions, generator=g)  # [S, N * repetitions]
    model = build_model(cfg, st, loss=loss, precision=precision)
    model.train()
    lb_w = torch.tensor([0.7 + 0.6 * s / max(cfg.num_subnetworks - 1, 1) for s in range(cfg.num_subnetworks)])
    model.loss_buffer.get_weights = lambda: lb_w  # fixed non-uniform weights on both sides
    out = model.training_step_with_perms(image.cuda(), label.cuda(), None if mask is None else mask.cuda(), perms.cuda())
    out["loss"].backward()
    ts32, ref = _oracle_run(cfg, st, image, label, mask, perms, lb_w, loss, torch.float32)
    _, ref64 = _oracle_run(cfg, st, image, label, mask, perms, lb_w, loss, torch.float64)
    half = cfg.out_channels // 2
    preds = out["preds"].view(N * repetitions, cfg.num_subnetworks, half, H, W).cpu()
    e_out = rel_err(preds, ref["out"][:, :, :half])
    e_loss = abs(out["loss"].item() - float(ref["total"])) / abs(float(ref["total"]))
    sd = model.state_dict()
    e_buf = max(rel_err(sd["model." + k].cpu(), v) for k, v in ts32.st.items() if "running" in k)
    grads = {k[len("model."):]: p.grad.detach().cpu().double() for k, p in model.named_parameters()}
    worst, dot, nh, nr, nd = ("", 0.0, 0.0), 0.0, 0.0, 0.0, 0.0
    n_tensors = arm_all = need_all = 0
    # The yardstick is what fp32 arithmetic itself does on this problem (fp32 oracle against the fp64 oracle): per tensor,
    # and over all tensors — training-mode BatchNorm over a small batch amplifies rounding into every gradient at once,
    # and WHICH tensor the fp32 oracle happens to get right depends on its thread count (reduction order)
    ks = [k for k in ref64["grads"] if not is_prebn_bias(k)]
    eo_all = (sum(float(((ref["grads"][k].double() - ref64["grads"][k]) ** 2).sum()) for k in ks)
              / sum(float((ref64["grads"][k] ** 2).sum()) for k in ks)) ** 0.5
    for k, g64 in ref64["grads"].items():
        if is_prebn_bias(k):
            continue  # mathematically zero; rounding noise on both sides
        eh = float((grads[k] - g64).norm() / g64.norm())
        eo = float((ref["grads"][k].double() - g64).norm() / g64.norm())
        if eh > worst[1]:
            worst = (k, eh, eo)
        floor = 5e-3 if small_net else 1e-3
        assert eh <= floor + 5.0 * max(eo, eo_all), (k, eh, eo, eo_all)
        n_tensors += 1
        arm_all += eo_all > eo               # the whole-gradient arm is the larger of the two for this tensor
        need_all += eh > floor + 5.0 * eo    # ... and the tensor would NOT have passed on its own fp32-oracle error
        dot += float((grads[k] * g64).sum())
        nh += float((grads[k] ** 2).sum())
        nr += float((g64 ** 2).sum())
        nd += float(((grads[k] - g64) ** 2).sum())
    cos, rel_l2 = dot / (nh * nr) ** 0.5, (nd / nr) ** 0.5
    report(f"[{precision}] out {e_out:.2e} loss {e_loss:.2e} bn-buffers {e_buf:.2e}; grads vs fp64: cos {cos:.7f} rel-L2 {rel_l2:.2e}; "
          f"worst tensor {worst[0]} hip {worst[1]:.2e} (fp32 oracle {worst[2]:.2e}; fp32 oracle over all tensors {eo_all:.2e}); "
          f"bound arms: eo_all > eo on {arm_all} of {n_tensors} tensors, {need_all} needed it to pass")
    assert e_out < TOL and e_loss < TOL and e_buf < TOL
    assert cos > (0.9995 if small_net else 0.9999) and rel_l2 < (3e-2 if small_net else 2e-2)
    return e_out, worst


@pytest.mark.parametrize("precision", PRECISIONS)
def test_cfg3_shape_vs_oracle(precision):
    """BASELINE config[2] geometry (2->1 ch, 256x256, S=2, fbc=30) at a batch the CPU oracle finishes in seconds."""
    e_out, worst = _oracle_vs_hip(O.NetConfig(2, 2, 2, 30), N=2, H=256, W=256, seed=5, precision=precision)
    report(f"cfg3-shape: out err {e_out:.2e}; worst grad {worst}")
    assert e_out < TOL


@pytest.mark.parametrize("precision", PRECISIONS)
def test_cfg2_shape_vs_oracle(precision):
    """BASELINE config[1] geometry (3->1 ch, S=2, fbc=21: channel counts 21/42/63/31 exercise every padding path)."""
    e_out, worst = _oracle_vs_hip(O.NetConfig(3, 2, 2, 21), N=2, H=128, W=128, seed=6, with_mask=True, precision=precision)
    report(f"cfg2-shape: out err {e_out:.2e}; worst grad {worst}")
    assert e_out < TOL


def test_s4_gaussian_vs_oracle():
    """S=4 head-width stress (BASELINE config[3] topology) with the Gaussian loss, small spatial size."""
    e_out, worst = _oracle_vs_hip(O.NetConfig(2, 2, 4, 6), N=3, H=48, W=64, seed=7, loss="gaussian_nll")
    report(f"S4: out err {e_out:.2e}; worst grad {worst}")
    assert e_out < TOL


@pytest.mark.parametrize("case", [
    # (Ci, Co, S, f, N, H, W, loss, mask)  — each exercises a path the BASELINE-shaped cases do not
    (3, 2, 2, 6, 2, 50, 70, "laplace_nll", True),     # odd sizes: floor pooling + zero F.pad in up-sampling, backward too
    (2, 2, 3, 5, 2, 100, 100, "laplace_nll", False),  # odd S, odd filter count (channel padding everywhere)
    (3, 2, 1, 8, 1, 64, 64, "laplace_nll", False),    # batch 1
    (3, 4, 2, 8, 2, 32, 48, "gaussian_nll", True),    # two targets (Co = 4) with a mask
    (1, 2, 1, 16, 2, 64, 64, "laplace_nll", False),   # single input channel, S = 1
    (2, 2, 2, 34, 1, 64, 96, "laplace_nll", False),   # widths 34 / 68 / 136 / 272 / 544: 48-wide tiles, 3 co tiles
], ids=lambda c: "-".join(map(str, c[:7])))
def test_more_geometries_vs_oracle(case):
    Ci, Co, S, f, N, H, W, loss, with_mask = case
    e_out, worst = _oracle_vs_hip(O.NetConfig(Ci, Co, S, f), N=N, H=H, W=W, seed=sum(case[:7]), loss=loss,
                                  with_mask=with_mask, small_net=f < 16)
    report(f"{case}: out err {e_out:.2e}; worst grad {worst}")


def test_batch_repetitions_and_input_repetition_vs_oracle():
    """apply_input_transform with batch_repetitions = 2 and input_repetition_probability = 0.5 (utils.py:27-48):
    the transformed batch is 2N wide and half of its rows show every subnetwork the same image."""
    e_out, worst = _oracle_vs_hip(O.NetConfig(3, 2, 2, 16), N=3, H=48, W=64, seed=11, with_mask=True, repetitions=2, irp=0.5,
                                  small_net=True)
    report(f"repetitions: out err {e_out:.2e}; worst grad {worst}")


def test_full_size_properties():
    """BASELINE-size run (cfg3: N=32, 256x256) checked through size-independent properties:
    bit-reproducibility of a training step, batch-permutation equivariance in eval mode,
    and loss == mean of the returned element-wise NLL."""
    cfg = O.NetConfig(2, 2, 2, 30)
    st = O.init_state(cfg, 3)
    model = build_model(cfg, st)
    g = torch.Generator().manual_seed(4)
    N = 32
    image = torch.rand(N, 2, 256, 256, generator=g).cuda()
    label = torch.rand(N, 1, 256, 256, generator=g).cuda()
    perms = O.draw_perms(N, 2, generator=g).cuda()
    model.train()
    outs = []
    for _ in range(2):
        model.load_state_dict({"model." + k: v for k, v in st.items()})
        model.loss_buffer.buffer.zero_()
        model.loss_buffer.index = 0
        model.zero_grad()
        o = model.training_step_with_perms(image, label, None, perms)
        o["loss"].backward()
        outs.append((o["loss"].item(), o["preds"].clone(), model.model.flat_gradients().clone()))
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    assert torch.isfinite(outs[0][2]).all()
    # loss vector == mean NLL of the returned predictions
    p1 = o["preds"].view(N, 2, 1, 256, 256)
    model.eval()
    with torch.no_grad():
        x5 = torch.stack([image[perms[s]] for s in range(2)], 1)
        a1, a2 = model(x5)
        pi = torch.randperm(N, generator=torch.Generator().manual_seed(1)).cuda()
        b1, b2 = model(x5[pi])
    assert rel_err(b1.cpu(), a1[pi].cpu()) < 1e-5 and rel_err(b2.cpu(), a2[pi].cpu()) < 1e-5
    assert p1.shape == a1.shape


def test_validation_step_matches_oracle():
    """validation_step (mimo_unet.py:146-183): repeat over S, per-subnetwork NLL, uncertainties,
    combined NLL — against the oracle's restatement, eval-mode BatchNorm."""
    fx = load_npz("mini_s2_step.npz")
    cfg = cfg_from_meta(fx["meta"])
    st = state_from(fx, "final/")
    st.pop("loss_buffer", None)
    model = build_model(cfg, st)
    model.eval()
    image, label, mask = (torch.from_numpy(fx[f"s0/{k}"]) for k in ("image", "label", "mask"))
    out = model.validation_step({"image": image.cuda(), "label": label.cuda(), "mask": mask.cuda()}, 0)
    S = cfg.num_subnetworks
    with torch.no_grad():
        o = O.mimo_unet_forward(cfg, st, O.repeat_subnetworks(image, S), training=False)
        p1, p2 = O.split_heads(o, cfg.out_channels)
        lab = O.repeat_subnetworks(label, S)
        val_loss = O.laplace_nll(p1, p2, lab, mask=O.repeat_subnetworks(mask, S), reduce_mean=False).mean(dim=(0, 2, 3, 4))
        mean, alea, epi = O.compute_uncertainties("laplace_nll", p1, p2)
        comb = O.calculate_dist_param("laplace_nll", torch.sqrt(alea + epi), log=True)
        val_comb = O.laplace_nll(p1.mean(dim=1), comb, label, mask=mask)
    assert abs(out["loss"].item() - float(val_loss.mean())) <= TOL * abs(float(val_loss.mean()))
    assert rel_err(out["preds"].cpu(), mean) < TOL
    assert rel_err(out["aleatoric_std_map"].cpu(), alea.sqrt()) < TOL
    assert rel_err(out["epistemic_std_map"].cpu(), epi.sqrt()) < TOL
    assert rel_err(out["err_map"].cpu(), mean - label) < TOL
    assert abs(model.logged["val_loss_combined"].item() - float(val_comb)) <= TOL * abs(float(val_comb))
    for s in range(S):
        assert abs(model.logged[f"val_loss_{s}"].item() - float(val_loss[s])) <= TOL * abs(float(val_loss[s]))
    assert set(out) == {"loss", "label", "preds", "aleatoric_std_map", "epistemic_std_map", "err_map", "mask"}
    # the fused epilogue's logged scalars: compute_regression_metrics (metrics.py:22-34) and the uncertainty means
    yh, y = mean.flatten().double(), label.flatten().double()
    ref = {"mae": (yh - y).abs().mean(), "mse": ((yh - y) ** 2).mean(), "rmse": ((yh - y) ** 2).mean().sqrt(),
           "r2": 1 - ((y - yh) ** 2).sum() / ((y - y.mean()) ** 2).sum()}
    for k, v in ref.items():
        assert abs(model.logged[f"metric_val/{k}"].item() - float(v)) <= TOL * max(abs(float(v)), 1e-3), k
    for k, t in (("aleatoric_std_mean", alea.sqrt()), ("epistemic_std_mean", epi.sqrt())):
        v = float(t.clip(0, 5).mean())
        assert abs(model.logged[f"metric_val/{k}"].item() - v) <= TOL * abs(v), k


def test_deep_ensemble_of_two_checkpoints(tmp_path):
    """EnsembleModule over two checkpoints (ensemble.py:42,95-113): predictions concatenated on the
    subnetwork axis, uncertainties over all of them; checkpoints written and re-loaded from disk."""
    from mimo.models.ensemble import EnsembleModule
    fx = load_npz("mini_s2_step.npz")
    cfg = cfg_from_meta(fx["meta"])
    states = [state_from(fx, "init/"), state_from(fx, "final/")]
    states[1].pop("loss_buffer", None)
    paths = []
    for i, st in enumerate(states):
        m = build_model(cfg, st).cpu()
        p = str(tmp_path / f"m{i}.ckpt")
        torch.save({"state_dict": m.state_dict(), "hyper_parameters": dict(m.hparams)}, p)
        paths.append(p)
    ens = EnsembleModule(paths).cuda()
    x = torch.from_numpy(fx["s1/image"])
    mean, alea, epi = ens(x.cuda())
    assert ens.num_subnetworks == 2 * cfg.num_subnetworks and not mean.is_cuda
    with torch.no_grad():
        outs = [O.mimo_unet_forward(cfg, st, O.repeat_subnetworks(x, cfg.num_subnetworks), training=False) for st in states]
        p1 = torch.cat([O.split_heads(o, cfg.out_channels)[0] for o in outs], 1)
        p2 = torch.cat([O.split_heads(o, cfg.out_channels)[1] for o in outs], 1)
        ref = O.compute_uncertainties("laplace_nll", p1, p2)
    for a, b in zip((mean, alea, epi), ref):
        assert rel_err(a, b) < TOL


def test_staged_backward_equals_monolithic_and_hook_ranges():
    """mimo_backward_stage 0..7 (data-parallel overlap path) == mimo_backward, and the gradient-ready hook is told
    disjoint ranges that walk the flat buffer from its tail (heads + decoders) to its head (encoders), each one
    final — bit-identical to the finished buffer — at the moment it is announced."""
    fx = load_npz("mini_s2_step.npz")
    cfg = cfg_from_meta(fx["meta"])
    image, label, perms = (torch.from_numpy(fx[f"s0/{k}"]).cuda() for k in ("image", "label", "perms"))
    grads, calls = [], []
    for staged in (False, True):
        model = build_model(cfg, state_from(fx, "init/"))
        model.train()
        if staged:
            model.model.grad_ready_hook = lambda flat, b, e: calls.append((b, e, flat[b:e].clone()))
        out = model.training_step_with_perms(image, label, None, perms)
        out["loss"].backward()
        grads.append(model.model.flat_gradients().clone())
    assert torch.equal(grads[0], grads[1])
    assert len(calls) == 8 and calls[0][1] == grads[0].numel() and calls[-1][0] == 0
    for i, (b, e, g) in enumerate(calls):
        assert b < e and torch.equal(g, grads[0][b:e])  # final when announced
        if i:
            assert e == calls[i - 1][0]  # adjacent, descending
    names = dict(model.model.named_parameters())
    enc_numel = sum(p.numel() for n, p in names.items() if n.startswith("encoder."))
    assert enc_numel <= calls[-1][1] <= enc_numel + 4 * len(names)  # encoder block (+16-byte alignment gaps)
    dec_numel = sum(p.numel() for n, p in names.items() if n.startswith("decoder."))
    assert dec_numel <= calls[0][1] - calls[0][0] <= dec_numel + 4 * len(names)
    # gradient accumulation under a hook: ranges are announced after the accumulation, with the accumulated values
    calls.clear()
    model.loss_buffer.get_weights = lambda: torch.ones(cfg.num_subnetworks, device="cuda")  # as in the first step
    out = model.training_step_with_perms(image, label, None, perms)
    out["loss"].backward()  # .grad still set: accumulates
    assert len(calls) == 8
    total = model.model.flat_gradients()
    for b, e, g in calls:
        assert torch.equal(g, total[b:e])
    assert torch.allclose(total, 2 * grads[0], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("geom", [(2, 2, 2, 30, 2, 256, 256), (3, 2, 2, 21, 3, 96, 80), (2, 2, 1, 12, 2, 64, 64)],
                         ids=lambda g: "-".join(map(str, g)))
def test_bn_relu_in_the_second_convolutions_loaders_is_bit_identical(geom, monkeypatch):
    """Round 4: in split16 the BatchNorm + ReLU between the two convolutions of a DoubleConv (components.py:24-25) is applied
    by the second convolution's loaders (forward: wide / 256-pixel kernels; weight gradient: wave-specialised kernel), and
    the activated tensor is never written.  Same arithmetic (relu(fma(z, scale, shift)) in fp32, then the 16-bit split) as
    the separate pass: predictions, loss, every gradient and the BatchNorm buffers are BIT-identical to MIMO_FUSE_BN_IN=0
    (cfg3 widths at 256 x 256: wide-kernel layers; cfg2 widths at an odd size: pairing tails and 32-channel weight-gradient
    tiles; a small net whose deep layers fall below 256 pixels and stay unfused)."""
    Ci, Co, S, f, N, H, W = geom
    cfg = O.NetConfig(Ci, Co, S, f)
    st = O.init_state(cfg, 77)
    g = torch.Generator().manual_seed(78)
    for k in st:  # non-trivial BatchNorm affine parameters (negative scales too)
        if (".double_conv.1." in k or ".double_conv.4." in k) and k.endswith("weight"):
            st[k] = torch.randn(st[k].shape, generator=g)
        if (".double_conv.1." in k or ".double_conv.4." in k) and k.endswith(".bias"):
            st[k] = 0.3 * torch.randn(st[k].shape, generator=g)
    image, label = torch.rand(N, Ci, H, W, generator=g).cuda(), torch.rand(N, 1, H, W, generator=g).cuda()
    perms = O.draw_perms(N, S, generator=g).cuda()
    res = []
    for flag in ("1", "0"):
        monkeypatch.setenv("MIMO_FUSE_BN_IN", flag)
        m = build_model(cfg, st)
        m.train()
        outs = []
        for _ in range(2):  # two steps: the second one runs on updated running statistics / reused buffers
            m.zero_grad()
            o = m.training_step_with_perms(image, label, None, perms)
            o["loss"].backward()
            outs.append((o["loss"].detach().clone(), o["preds"].clone(), m.model.flat_gradients().clone()))
        m.eval()  # eval mode with a graph (FGSM): the separate pass stays, the weight gradient still reads z
        x5 = torch.stack([image[perms[s]] for s in range(S)], 1).requires_grad_(True)
        p1, p2 = m(x5)
        m.zero_grad()
        (p1.mean() + p2.mean()).backward()
        outs.append((p1.detach().clone(), x5.grad.clone(), m.model.flat_gradients().clone()))
        res.append((outs, {k: v.clone() for k, v in m.state_dict().items()}))
    (a, sa), (b, sb) = res
    for ta, tb in zip(a, b):
        for x, y in zip(ta, tb):
            assert torch.equal(x, y)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_weights_beyond_the_old_fixed_fp16_scale_stay_finite_and_close():
    """VERDICT r4 missing 4 / item 7: the reference's fp32 Conv2d (components.py:23,26) is finite for any fp32 weight; the
    split16 forward carried its weights as fp16 pairs x 2^8, so one |w| >= 256 turned its output channel into inf / NaN
    (detected, not handled).  Round 5: the scale of a layer's fp16 image follows its largest |w| from 128 up (w16_scale:
    the power of two that puts max |w| into [2^13, 2^14), applied by the packer and removed in the convolution's epilogue;
    ordinary layers keep 2^8 and their bits).  A checkpoint with weights of 300 and -4000 in layers of every forward kernel
    family (256-pixel wave-specialised, the small-image kernel of the deep layers; the forced-wide variant run covers
    conv_wide.hip): training step and eval forward finite and within the usual tolerances of the fp32 oracle."""
    cfg = O.NetConfig(2, 2, 2, 30)
    st = O.init_state(cfg, 5)
    g = torch.Generator().manual_seed(6)
    for name in ("encoder.in_convs.0.double_conv.3.weight", "encoder.down1s.1.conv.double_conv.0.weight",
                 "core.down2.conv.double_conv.3.weight", "core.down4.conv.double_conv.0.weight",
                 "core.up2.conv.double_conv.0.weight", "decoder.up4s.1.conv.double_conv.3.weight"):
        w = st[name].view(-1)
        idx = torch.randint(0, w.numel(), (4,), generator=g)
        w[idx] = torch.tensor([300.0, -4000.0, 130.0, 257.0])
    N, H, W = 4, 64, 64
    image, label = torch.rand(N, 2, H, W, generator=g), torch.rand(N, 1, H, W, generator=g)
    perms = O.draw_perms(N, 2, generator=g)
    model = build_model(cfg, st)
    model.train()
    out = model.training_step_with_perms(image.cuda(), label.cuda(), None, perms.cuda())
    out["loss"].backward()
    lb_w = torch.ones(2)  # first step: loss-buffer weights are exactly 1
    ts, ref = _oracle_run(cfg, st, image, label, None, perms, lb_w, "laplace_nll", torch.float32)
    _, ref64 = _oracle_run(cfg, st, image, label, None, perms, lb_w, "laplace_nll", torch.float64)
    preds = out["preds"].view(N, 2, 1, H, W).cpu()
    assert torch.isfinite(preds).all() and torch.isfinite(model.model.flat_gradients()).all()
    e_out = rel_err(preds, ref["out"][:, :, :1])
    e_loss = abs(float(out["loss"].detach()) - float(ref["total"])) / abs(float(ref["total"]))
    # gradients: weights of thousands make the backward ill-conditioned (the fp32 oracle itself sits several 1e-3 from the
    # fp64 one on the early layers) — anchored on fp64 like _oracle_vs_hip: within 2e-3 + 5 x the fp32 oracle's own error
    worst = ("", 0.0, 0.0)
    for k, g64 in ref64["grads"].items():
        if is_prebn_bias(k):
            continue
        gh = dict(model.named_parameters())["model." + k].grad.detach().cpu().double()
        eh = float((gh - g64).norm() / g64.norm())
        eo = float((ref["grads"][k].double() - g64).norm() / g64.norm())
        if eh > worst[1]:
            worst = (k, eh, eo)
        assert eh <= 2e-3 + 5.0 * eo, (k, eh, eo)
    model.eval()
    with torch.no_grad():
        x5 = torch.stack([image[perms[s]] for s in range(2)], 1)
        p1, p2 = model(x5.cuda())
        ev = O.mimo_unet_forward(cfg, ts.st, x5, training=False)
    e_eval = rel_err(torch.cat([p1, p2], dim=2).cpu(), ev)
    assert model.model.numerics_status() == 0
    report(f"weights of 300 / -4000 in six layers: train out {e_out:.2e} loss {e_loss:.2e} worst grad {worst}; eval out {e_eval:.2e}")
    assert e_out < TOL and e_loss < TOL and e_eval < TOL


@pytest.mark.parametrize("kind", ["dropout2d", "final"])
def test_caller_masks_on_blocks_built_without_dropout_are_honoured_in_both_directions(kind, monkeypatch):
    """`mimo_forward_args.drop_masks[i]` / `elem_masks[j]` are honoured on every block, also on one the
    plan was built with rate 0 for — there the output activation is elided by construction (its readers apply BatchNorm +
    ReLU to z), and round 4's forward dropped the multiplier while the backward still applied it.  Now such a call
    materialises the tensor: the step equals the oracle's step with the same multipliers (reference: Dropout2d /
    nn.Dropout after the block, components.py:29, model.py:277-281) and is bit-identical to MIMO_FUSE_BN_IN=0."""
    N, H, W = 2, 64, 64
    S = 2
    ocfg = (O.NetConfig(2, 2, S, 16, core_dropout_rate=0.5, decoder_dropout_rate=0.5) if kind == "dropout2d" else
            O.NetConfig(2, 2, S, 16, final_dropout_rate=0.5))
    cfg0 = O.NetConfig(2, 2, S, 16)  # what the HIP plan is built for: no dropout anywhere
    st = O.init_state(cfg0, 31)
    g = torch.Generator().manual_seed(32)
    image, label = torch.rand(N, 2, H, W, generator=g), torch.rand(N, 1, H, W, generator=g)
    perms = O.draw_perms(N, S, generator=g)
    specs = O.double_conv_specs(cfg0)  # (prefix, cin, mid, cout) in the engine's order
    omasks, hmask, hemask = {}, {}, {}
    if kind == "dropout2d":
        for j, (prefix, _, _, cout) in enumerate(specs):
            if prefix.startswith(("core.up", "decoder.up4s")):  # the blocks whose outputs are elided
                m = (torch.rand(N, cout, generator=g) > 0.5).float() * 2.0
                omasks[prefix], hmask[j] = m, m
            elif prefix.startswith("core."):  # the oracle's rate applies to the whole core: all-ones there
                omasks[prefix] = torch.ones(N, cout)
    else:
        for s_ in range(S):
            m = (torch.rand(N, 16, H, W, generator=g) > 0.5).float() * 2.0
            omasks[f"decoder.final_dropouts.{s_}"], hemask[f"final{s_}"] = m, m
    ts = O.TrainState(cfg=ocfg, st={k: v.clone() for k, v in st.items()}, loss_buffer=O.LossBuffer(S, 0.3, 10))
    ref = O.train_step(ts, image, label, None, perms, masks=omasks, apply_optimizer=False)
    res = []
    for flag in ("1", "0"):
        monkeypatch.setenv("MIMO_FUSE_BN_IN", flag)
        m = build_model(cfg0, st)
        m.train()
        m.model.mask_override = hmask or None
        m.model.elem_mask_override = hemask or None
        o = m.training_step_with_perms(image.cuda(), label.cuda(), None, perms.cuda())
        o["loss"].backward()
        grads = {k[len("model."):]: p.grad.detach().cpu() for k, p in m.named_parameters()}
        res.append((o["loss"].detach().cpu(), o["preds"].cpu(), m.model.flat_gradients().clone().cpu()))
        e_out = rel_err(o["preds"].view(N, S, 1, H, W).cpu(), ref["out"][:, :, :1])
        e_loss = abs(float(o["loss"]) - float(ref["total"])) / abs(float(ref["total"]))
        # a 16-channel net at 64 x 64 with half of several blocks' channels dropped: single ReLU / max-pool flips move a
        # tensor by up to ~5e-3 of its scale (observed 5.2e-3 on one BatchNorm bias); what this test is after — a multiplier
        # applied in one direction only — moves the masked blocks' gradients by O(1)
        worst = check_grads(grads, ref["grads"], tol=2e-2)
        report(f"{kind} multipliers on rate-0 blocks, MIMO_FUSE_BN_IN={flag}: out {e_out:.1e} loss {e_loss:.1e} worst grad {worst}")
        assert e_out < TOL and e_loss < TOL
    for x, y in zip(*res):
        assert torch.equal(x, y)


@pytest.mark.parametrize("geom", [(2, 2, 2, 30, 2, 256, 256, 0.0), (3, 2, 3, 10, 3, 100, 100, 0.2), (1, 2, 1, 8, 2, 50, 70, 0.0)],
                         ids=lambda g: "-".join(map(str, g)))
def test_pool_and_head_gradients_formed_by_the_batchnorm_backward(geom, monkeypatch):
    """Round 4: the gradient arriving at a pooled (and skip-connected) encoder tensor — MaxPool2d backward of the Down block's
    data gradient + the skip half of torch.cat's backward (components.py:48,118) — and the one arriving at the head's input
    (1x1 conv + NLL backward, components.py:126, losses.py:151-160) are formed by the two BatchNorm-backward passes of the
    producing convolution; pool_bwd / head_bwd are not launched and those gradient tensors are never written.
    Each element's arriving gradient has the same bits as what the separate kernel wrote (same operands, same order), but the
    launches differ — one thread owns a 2x2 window of a pooled tensor, the fused reductions run on one workgroup per CU — so
    the per-channel sums of BatchNorm's backward (and the head's own weight / bias gradient) group their terms differently:
    against MIMO_FUSE_BWD_SRC=0 the forward quantities and BatchNorm buffers are bit-identical and every gradient agrees to
    fp32 summation-order noise (bound 1e-4 of each tensor's scale — a tenth of the parity tolerance; observed 1.5e-5 ... 1.9e-5
    on the worst tensor of each case, profiles/r04/parity_errors.txt); =3 is the head alone, =1 both.
    Odd sizes (100 -> 50 -> 25 -> 12: rows / columns outside every window), Dropout2d masks on the encoder blocks, a masked
    loss, S = 1 and S = 3, and the eval-mode backward (FGSM)."""
    Ci, Co, S, f, N, H, W, p_enc = geom
    cfg = O.NetConfig(Ci, Co, S, f, encoder_dropout_rate=p_enc)
    st = O.init_state(cfg, 91)
    g = torch.Generator().manual_seed(92)
    for k in st:
        if (".double_conv.1." in k or ".double_conv.4." in k) and k.endswith("weight"):
            st[k] = torch.randn(st[k].shape, generator=g)
        if (".double_conv.1." in k or ".double_conv.4." in k) and k.endswith(".bias"):
            st[k] = 0.3 * torch.randn(st[k].shape, generator=g)
    image, label = torch.rand(N, Ci, H, W, generator=g).cuda(), torch.rand(N, 1, H, W, generator=g).cuda()
    mask = (torch.rand(N, 1, H, W, generator=g) > 0.3).float().cuda() if S == 3 else None
    perms = O.draw_perms(N, S, generator=g).cuda()
    res = {}
    for flag in ("0", "3", "1"):
        monkeypatch.setenv("MIMO_FUSE_BWD_SRC", flag)
        torch.manual_seed(93)
        m = build_model(cfg, st, dropout=(p_enc, 0.0, 0.0))
        m.train()
        outs = []
        for _ in range(2):
            m.zero_grad()
            o = m.training_step_with_perms(image, label, mask, perms)
            o["loss"].backward()
            outs.append((o["loss"].detach().clone(), o["preds"].clone(), {n: p.grad.clone() for n, p in m.model.named_parameters()}))
        m.eval()
        x5 = torch.stack([image[perms[s]] for s in range(S)], 1).requires_grad_(True)
        p1, p2 = m(x5)
        m.zero_grad()
        (p1.mean() + p2.mean()).backward()
        outs.append((p1.detach().clone(), x5.grad.clone(), {n: p.grad.clone() for n, p in m.model.named_parameters()}))
        res[flag] = (outs, {k: v.clone() for k, v in m.state_dict().items()})

    def close(x, y, scale, tol):
        return float((x - y).abs().max()) <= tol * (float(scale.abs().max()) + 1e-30)

    (b, sb) = res["0"]
    worst = 0.0
    for flag in ("3", "1"):
        a, sa = res[flag]
        for ta, tb in zip(a, b):
            assert torch.equal(ta[0], tb[0])  # forward quantities: untouched
            for x, y, n in [(ta[1], tb[1], "second output")] + [(ta[2][n], tb[2][n], n) for n in ta[2]]:
                # (pre-BatchNorm conv biases: a mathematically zero gradient, judged on the scale of their weight gradient)
                scale = tb[2][n[:-4] + "weight"] if is_prebn_bias(n) else y
                worst = max(worst, float((x - y).abs().max()) / (float(scale.abs().max()) + 1e-30))
                assert close(x, y, scale, 1e-4), (flag, n)
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k
    report(f"fused gradient sources {geom}: worst deviation from the separate kernels {worst:.2e} of the tensor's scale (bound 1e-4)")


def test_plan_create_rejects_block_variants_the_reference_does_not_have():
    """mimo_config.norm_kind / act_kind / up_kind: 0 = BatchNorm2d + ReLU + bilinear align_corners, the reference's
    blocks and the only implemented ones; anything else is MIMO_ERR_INVALID with a message, not silently BatchNorm."""
    import ctypes as C
    from mimo_unet_amd import _lib as L
    lib = L.load()

    def create(**kw):
        cfg = L.MimoConfig(2, 2, 1, 4, 1, 32, 32, 0.0, 0.0, 0.0, 1e-5, 0.1, 0, 1e-5, 1e3, 0, 1, 0, 0.0, 0.0, 0, 0, 0)
        for k, v in kw.items():
            setattr(cfg, k, v)
        h = C.c_void_p()
        rc = lib.mimo_plan_create(C.byref(cfg), C.byref(h))
        if rc == 0:
            lib.mimo_plan_destroy(h)
        return rc, lib.mimo_last_error().decode()

    assert create()[0] == 0
    for field in ("norm_kind", "act_kind", "up_kind"):
        rc, msg = create(**{field: 1})
        assert rc == -1 and "not implemented" in msg, (field, rc, msg)


def test_training_forward_always_uses_the_current_parameters():
    """Every training forward repacks the convolution weights from the flat parameter buffer, so a parameter write torch
    cannot see (`p.data.mul_()`, a raw-pointer writer) between `optimizer.step()` and the next training forward takes
    effect without `mark_parameters_changed()`; only the cached EVAL weights need that call (the round-3
    prepack had silently widened that contract to training and was removed)."""
    fx = load_npz("mini_s2_step.npz")
    cfg = cfg_from_meta(fx["meta"])
    image, label, perms = (torch.from_numpy(fx[f"s0/{k}"]).cuda() for k in ("image", "label", "perms"))
    state0 = state_from(fx, "init/")
    model = build_model(cfg, state0)
    model.train()
    opt = model.configure_optimizers()["optimizer"]
    out = model.training_step_with_perms(image, label, None, perms)
    out["loss"].backward()
    opt.step()
    with torch.no_grad():
        for p in model.parameters():
            p.data.mul_(0.5)  # invisible to torch's version counters; no mark_parameters_changed()
    state1 = {k[len("model."):]: v.detach().clone() for k, v in model.state_dict().items()}
    opt.zero_grad()
    out2 = model.training_step_with_perms(image, label, None, perms)
    fresh = build_model(cfg, state1)
    fresh.train()
    ref = fresh.training_step_with_perms(image, label, None, perms)
    assert torch.equal(out2["preds"], ref["preds"])  # (the weighted loss differs: the loss buffer holds step 1)


@pytest.mark.parametrize("name", ["mini_s2_step.npz", "cfg1_step.npz"])
def test_side_stream_weight_gradients_match(name, monkeypatch):
    """MIMO_WGRAD_STREAM=1 (weight gradients on a side stream with ping-pong dz buffers, plain and staged
    backward) produces bit-identical gradients: same kernels, only the stream they run on changes."""
    fx = load_npz(name)
    cfg = cfg_from_meta(fx["meta"])
    image, label, perms = (torch.from_numpy(fx[f"s0/{k}"]).cuda() for k in ("image", "label", "perms"))
    grads = []
    for mode in ("0", "1", "1-staged"):
        monkeypatch.setenv("MIMO_WGRAD_STREAM", mode[0])
        model = build_model(cfg, state_from(fx, "init/"))  # the variable is read when the plan is created
        model.train()
        if mode.endswith("staged"):
            model.model.grad_ready_hook = lambda flat, b, e: None
        for _ in range(2):  # second iteration: buffers of the first one are reused across the stream join
            model.zero_grad()
            out = model.training_step_with_perms(image, label, None, perms)
            out["loss"].backward()
        torch.cuda.synchronize()
        grads.append(model.model.flat_gradients().clone())
    assert all(torch.equal(grads[0], g) for g in grads[1:])


def test_side_stream_weight_gradients_match_on_a_small_odd_geometry(monkeypatch):
    """The same on the sequence where a race showed at the end of round 5 (S = 3, fbc = 10, 100 x 100, batch 3, masked loss; two
    training iterations and an eval-mode backward, the head's gradient source fused): the side stream's weight-gradient
    REDUCTION read the max |dz| slots of a dz buffer whose release event had been recorded in front of it, while the main
    stream was already through the BatchNorm backward of the layer after next.  Three repetitions on the side stream, every
    gradient bit-identical to the same sequence with everything on one stream."""
    cfg = O.NetConfig(2, 2, 3, 10, encoder_dropout_rate=0.2)
    st = O.init_state(cfg, 91)
    g = torch.Generator().manual_seed(92)
    for k in st:
        if (".double_conv.1." in k or ".double_conv.4." in k) and k.endswith("weight"):
            st[k] = torch.randn(st[k].shape, generator=g)
    image, label = torch.rand(3, 2, 100, 100, generator=g).cuda(), torch.rand(3, 1, 100, 100, generator=g).cuda()
    mask = (torch.rand(3, 1, 100, 100, generator=g) > 0.3).float().cuda()
    perms = O.draw_perms(3, 3, generator=g).cuda()
    monkeypatch.setenv("MIMO_FUSE_BWD_SRC", "3")
    runs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MIMO_WGRAD_STREAM", mode)
        grads = []
        for rep in range(3):
            torch.manual_seed(93)
            model = build_model(cfg, st, dropout=(0.2, 0.0, 0.0))
            model.train()
            for _ in range(2):
                model.zero_grad()
                model.training_step_with_perms(image, label, mask, perms)["loss"].backward()
                grads.append(model.model.flat_gradients().clone())
            model.eval()
            x5 = torch.stack([image[perms[s]] for s in range(3)], 1).requires_grad_(True)
            p1, p2 = model(x5)
            model.zero_grad()
            (p1.mean() + p2.mean()).backward()
            grads.append(model.model.flat_gradients().clone())
        torch.cuda.synchronize()
        runs[mode] = grads
    for i, (a, b) in enumerate(zip(runs["0"], runs["1"])):
        assert torch.equal(a, b), f"gradient set {i}: max difference {float((a - b).abs().max()):.3e}"


def test_inference_path_matches_eval_forward_and_tracks_parameter_changes():
    """Eval mode under torch.no_grad() takes the inference path (BatchNorm + ReLU + Dropout2d multipliers
    in the convolution epilogue, packed weights cached on the parameter version): bit-identical to the
    grad-enabled eval forward, and never stale after an optimiser step, a training forward (running
    statistics) or load_state_dict."""
    fx = load_npz("mc_dropout.npz")
    Ci, Co, S, f, N, H, W, passes = (int(v) for v in fx["meta"])
    p = float(fx["p"])
    cfg = O.NetConfig(Ci, Co, S, f)
    state = state_from(fx, "state/")
    model = build_model(cfg, state, dropout=(p, p, p))
    model.eval()
    x = repeat_sub(torch.from_numpy(fx["x"]).cuda(), S)
    prefixes = [s[0] for s in O.double_conv_specs(cfg)]
    model.model.mask_override = {j: torch.from_numpy(fx[f"pass0/mask{j}"]) for j in range(len(prefixes))}

    def both():
        a1, a2 = model(x)                      # autograd-capable eval forward (z kept, separate BN/ReLU pass)
        with torch.no_grad():
            b1, b2 = model(x)                  # inference path
            c1, c2 = model(x)                  # again: cached weights / graph replay
        assert torch.equal(a1, b1) and torch.equal(a2, b2) and torch.equal(b1, c1) and torch.equal(b2, c2)
        return b1.clone()

    r0 = both()
    assert rel_err(r0.cpu(), fx["p1"][:, :S]) < TOL                      # pass 0 of the golden ensemble
    # optimiser step through the raw-pointer kernel
    model.train()
    opt = model.configure_optimizers()["optimizer"]
    y = torch.rand(N, S, Co // 2, H, W, device="cuda")
    p1, p2 = model(x)
    model.loss_fn.forward(p1, p2, y).backward()
    opt.step()
    model.eval()
    r1 = both()
    assert not torch.equal(r0, r1)
    fresh = build_model(cfg, {k[len("model."):]: v for k, v in model.state_dict().items()}, dropout=(p, p, p))
    fresh.eval()
    fresh.model.mask_override = model.model.mask_override
    with torch.no_grad():
        f1, _ = fresh(x)
    assert torch.equal(f1, r1)
    # load_state_dict back to the original weights
    model.load_state_dict({"model." + k: v for k, v in state.items()})
    assert torch.equal(both(), r0)
    # backward after an inference forward is refused, after a grad-enabled eval forward it works
    xg = x.clone().requires_grad_(True)
    q1, q2 = model(xg)
    (q1.sum() + q2.sum()).backward()
    assert torch.isfinite(xg.grad).all()


def test_inference_cache_is_invalidated_by_load_state_dict_alone():
    """no_grad forward, load_state_dict, no_grad forward — with NO grad-enabled call in between (which would reset
    the cache by itself): the second forward must use the new weights, also for in-place parameter writes torch
    can see (p.copy_) and after mark_parameters_changed() for raw writes through .data."""
    fx = load_npz("mini_s2_step.npz")
    cfg = cfg_from_meta(fx["meta"])
    s_a, s_b = state_from(fx, "init/"), state_from(fx, "final/")
    s_b.pop("loss_buffer", None)
    x = repeat_sub(torch.from_numpy(fx["s0/image"]).cuda(), cfg.num_subnetworks)
    want = []
    for st in (s_a, s_b):
        m = build_model(cfg, st)
        m.eval()
        with torch.no_grad():
            want.append(m(x)[0].clone())
    assert not torch.equal(want[0], want[1])
    model = build_model(cfg, s_a)
    model.eval()
    with torch.no_grad():
        assert torch.equal(model(x)[0], want[0])
        assert torch.equal(model(x)[0], want[0])          # served from the cache / graph replay
        model.load_state_dict({"model." + k: v for k, v in s_b.items()})
        assert torch.equal(model(x)[0], want[1])
        assert torch.equal(model(x)[0], want[1])
        for k, p in model.model.named_parameters():      # in-place copies torch's version counters see
            p.copy_(s_a[k].cuda())
        for k, b in model.model.named_buffers():
            b.copy_(s_a[k].cuda())
        assert torch.equal(model(x)[0], want[0])
        for k, p in model.model.named_parameters():      # raw writes: invisible to torch, announced by hand
            p.data.copy_(s_b[k].cuda())
        for k, b in model.model.named_buffers():
            b.data.copy_(s_b[k].cuda())
        model.model.mark_parameters_changed()
        assert torch.equal(model(x)[0], want[1])


def test_flat_adam_checkpoint_resume_on_device(tmp_path):
    """Optimiser state saved, re-loaded with map_location="cpu" (what Lightning's checkpoint IO delivers) and
    stepped: identical parameters to the uninterrupted run — the moments are moved, not silently re-zeroed."""
    fx = load_npz("mini_s2_step.npz")
    cfg = cfg_from_meta(fx["meta"])
    image, label, perms = (torch.from_numpy(fx[f"s0/{k}"]).cuda() for k in ("image", "label", "perms"))

    def one_step(m, opt):
        opt.zero_grad()
        m.training_step_with_perms(image, label, None, perms)["loss"].backward()
        opt.step()

    model = build_model(cfg, state_from(fx, "init/"))
    model.train()
    opt = model.configure_optimizers()["optimizer"]
    one_step(model, opt)
    one_step(model, opt)
    path = str(tmp_path / "ck.pt")
    torch.save({"state_dict": model.state_dict(), "optimizer": opt.state_dict(),
                "loss_buffer": (model.loss_buffer.buffer.clone(), model.loss_buffer.index)}, path)
    one_step(model, opt)
    want = model.model.flat_parameters().clone()
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert not ck["optimizer"]["flat"]["exp_avg"].is_cuda
    resumed = build_model(cfg, state_from(fx, "init/"))
    resumed.load_state_dict(ck["state_dict"])
    resumed.train()
    # (on the device, where the uninterrupted run keeps it: a CPU softmax of the same ring differs in the last bit)
    resumed.loss_buffer.buffer = ck["loss_buffer"][0].cuda()
    resumed.loss_buffer.index = ck["loss_buffer"][1]
    opt2 = resumed.configure_optimizers()["optimizer"]
    opt2.load_state_dict(ck["optimizer"])
    one_step(resumed, opt2)
    assert opt2._step == 3 and torch.equal(resumed.model.flat_parameters(), want)


def repeat_sub(x, S):
    return x[:, None].repeat(1, S, 1, 1, 1).contiguous()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_evidential_model_golden(precision):
    """EvidentialUnetModel (evidential_unet.py:13-145) on the HIP backbone: NIG outputs, per-pixel loss,
    variances, input gradient and every parameter gradient against the reference; validation_step keys."""
    from mimo.models.evidential_unet import EvidentialUnetModel
    fx = load_npz("evidential.npz")
    Ci, Co, S, f, N, H, W = (int(v) for v in fx["meta"])
    m = EvidentialUnetModel(in_channels=Ci, out_channels=Co, filter_base_count=f, center_dropout_rate=0.0,
                            final_dropout_rate=0.0, encoder_dropout_rate=0.0, core_dropout_rate=0.0,
                            decoder_dropout_rate=0.0, weight_decay=0.0, learning_rate=1e-3, seed=0)
    m.load_state_dict({"model." + k: v for k, v in state_from(fx, "init/").items()})
    m.model.set_precision(precision)
    m = m.cuda()
    m.train()
    x = torch.from_numpy(fx["x"]).cuda().requires_grad_(True)
    y, mask = torch.from_numpy(fx["y"]).cuda(), torch.from_numpy(fx["mask"]).cuda()
    out = m.training_step({"image": x, "label": y, "mask": mask}, 0)
    out["loss"].backward()
    np.testing.assert_allclose(out["loss"].item(), fx["loss"].mean(), rtol=TOL)
    assert rel_err(out["preds"][:, 0].detach().cpu(), fx["ev"][:, 0]) < TOL
    assert rel_err(out["aleatoric_std_map"][:, 0].detach().cpu() ** 2, fx["aleatoric_var"]) < TOL
    assert set(out) == {"loss", "label", "preds", "aleatoric_std_map", "err_map", "mask"}
    e_dx = rel_err(x.grad.cpu(), fx["dx"])
    grads = {k[len("model."):]: p.grad.detach().cpu() for k, p in m.named_parameters()}
    worst = check_grads(grads, {k[len("grad/"):]: v for k, v in fx.items() if k.startswith("grad/")}, tol=golden_grad_tol(precision))
    report(f"evidential [{precision}]: dx {e_dx:.2e} worst grad {worst[1]:.2e} at {worst[0]}")
    assert e_dx < TOL
    m.eval()
    v = m.validation_step({"image": x.detach(), "label": y, "mask": mask}, 0)
    assert set(v) == {"loss", "label", "preds", "aleatoric_std_map", "epistemic_std_map", "err_map", "mask"}
    assert all(torch.isfinite(t).all() for t in v.values() if isinstance(t, torch.Tensor))
    opt = m.configure_optimizers()["optimizer"]
    opt.step()  # fused Adam over the flat buffer


@pytest.mark.parametrize("name", ["cfg1_step.npz", "mini_s2_step.npz"])
def test_bf16_precision_mode(name):
    """MIMO_PREC_BF16 (bf16 MFMA operands, fp32 accumulate / storage — SURVEY §8a row U, BASELINE config 4's
    arithmetic).  The kernels themselves are checked to 2e-6 against rounded-operand references in
    test_ops_gpu.py.  Here: (a) eval-mode forward against the oracle with the engine's rounding policy
    (`O.conv_operands("bf16")`): same arithmetic up to fp32 summation order, but a 1e-7 difference can move an
    activation across a bf16 rounding boundary (4e-3 jump), so the agreement is ~3e-5, not 1e-7; (b) a
    training step against the fp32 reference golden and the bf16 oracle at bf16 tolerance — training-mode
    BatchNorm on these tiny batches amplifies the rounding-boundary noise ~400x, which bounds what any
    bf16 implementation can reproduce."""
    fx = load_npz(name)
    cfg = cfg_from_meta(fx["meta"])
    lr, wd = float(fx["lr"]), float(fx["wd"])
    image, label, perms = (torch.from_numpy(fx[f"s0/{k}"]) for k in ("image", "label", "perms"))
    mask = torch.from_numpy(fx["s0/mask"]) if "s0/mask" in fx else None
    S, N, half = cfg.num_subnetworks, image.shape[0], cfg.out_channels // 2
    model = build_model(cfg, state_from(fx, "init/"), loss=str(fx["loss_kind"]), lr=lr, wd=wd, T=float(fx["temperature"]),
                        precision="bf16")
    # (a) eval forward
    x = torch.stack([image[perms[s]] for s in range(S)], dim=1)
    model.eval()
    with torch.no_grad():
        p1, p2 = model(x.cuda())
        with O.conv_operands("bf16"):
            o16 = O.mimo_unet_forward(cfg, state_from(fx, "init/"), x, training=False)
        o32 = O.mimo_unet_forward(cfg, state_from(fx, "init/"), x, training=False)
    hip = torch.cat([p1, p2], dim=2).cpu()
    e16, e32 = rel_err(hip, o16), rel_err(hip, o32)
    # (b) training step
    ts = O.TrainState(cfg=cfg, st=state_from(fx, "init/"), loss_kind=str(fx["loss_kind"]), lr=lr, weight_decay=wd,
                      loss_buffer=O.LossBuffer(S, float(fx["temperature"]), 10))
    with O.conv_operands("bf16"):
        ref = O.train_step(ts, image, label, mask, perms, apply_optimizer=False)
    model.train()
    out = model.training_step_with_perms(image.cuda(), label.cuda(), None if mask is None else mask.cuda(), perms.cuda())
    out["loss"].backward()
    preds = out["preds"].view(N, S, half, *image.shape[-2:]).cpu()
    t16, t32 = rel_err(preds, ref["out"][:, :, :half]), rel_err(preds, torch.from_numpy(fx["s0/out"])[:, :, :half])
    dot = n1 = n2 = 0.0
    for k, p in model.named_parameters():
        if is_prebn_bias(k):
            continue
        g, r = p.grad.detach().cpu().double(), torch.from_numpy(fx["s0/grad/" + k[len("model."):]]).double()
        dot, n1, n2 = dot + float((g * r).sum()), n1 + float((g * g).sum()), n2 + float((r * r).sum())
    cos = dot / (n1 * n2) ** 0.5
    report(f"bf16 {name}: eval fwd vs bf16 oracle {e16:.2e} (vs fp32 {e32:.2e}); train out vs bf16 oracle {t16:.2e}, "
          f"vs fp32 golden {t32:.2e}; gradient cosine vs fp32 golden {cos:.4f}")
    # observed (profiles/r02/parity_errors.txt): e16 3.0e-5 / 1.8e-5, e32 2.9e-4 / 9.0e-5, t16 1.3e-2 / 2.3e-2,
    # t32 3.7e-2 / 4.3e-2, 1 - cosine 1.5e-2 / 4.7e-2 — bounds = about 5x the larger observation
    assert e16 < 1.5e-4 and e16 < e32 and e32 < 1.5e-3
    assert t16 < 1e-1 and t32 < 2e-1 and cos > 0.9
    np.testing.assert_allclose(out["loss"].item(), float(ref["total"]), rtol=2e-2, atol=2e-3)
    np.testing.assert_allclose(out["loss"].item(), fx["s0/total"], rtol=5e-2, atol=5e-3)


def test_evidential_head_loss_kernel_against_reference_golden_and_torch_autograd():
    """mimo_evidential_forward / _backward (softplus heads + evidential loss in one pass each): NIG parameters and
    per-pixel loss against the reference's golden values, the analytic gradient w.r.t. the logits (incl. the digamma
    term) against torch autograd through the host loss class, with both upstream gradients (loss map and parameters)."""
    from mimo.losses import EvidentialLoss
    from mimo_unet_amd.engine import evidential_head_loss
    fx = load_npz("evidential.npz")
    ev_ref, y, mask = (torch.from_numpy(fx[k]) for k in ("ev", "y", "mask"))
    # logits that reproduce the golden NIG parameters: inverse softplus
    inv = lambda t: torch.where(t > 20, t, torch.log(torch.expm1(t.double())).float())
    logits = torch.stack([ev_ref[:, 0], inv(ev_ref[:, 1]), inv(ev_ref[:, 2] - 1), inv(ev_ref[:, 3])], dim=1)
    lg = logits.cuda().requires_grad_(True)
    ev, loss = evidential_head_loss(lg, y.cuda(), mask.cuda())
    e_ev, e_loss = rel_err(ev.detach().cpu(), ev_ref), rel_err(loss.detach().cpu(), fx["loss"])
    w_loss = torch.rand(loss.shape, generator=torch.Generator().manual_seed(1)).cuda()
    w_ev = torch.rand(ev.shape, generator=torch.Generator().manual_seed(2)).cuda()
    ((loss * w_loss).sum() + (ev * w_ev).sum()).backward()
    # the same through torch's autograd (fp64 for the reference derivative)
    lt = logits.double().requires_grad_(True)
    mu, lv, la, lb = torch.unbind(lt, dim=1)
    sp = torch.nn.functional.softplus
    ev_t = torch.stack([mu, sp(lv), sp(la) + 1, sp(lb)], dim=1)
    loss_t = EvidentialLoss(coeff=1.0)(ev_t, y.double(), mask=mask.double())
    ((loss_t * w_loss.cpu().double()).sum() + (ev_t * w_ev.cpu().double()).sum()).backward()
    e_grad = rel_err(lg.grad.cpu(), lt.grad)
    report(f"evidential kernels: NIG parameters {e_ev:.2e}, loss map {e_loss:.2e}, dlogits vs fp64 autograd {e_grad:.2e}")
    assert e_ev < 1e-5 and e_loss < 1e-4 and e_grad < 1e-4
    # heads only (inference): no label
    ev2, none = evidential_head_loss(lg.detach())
    assert none is None and torch.equal(ev2, ev.detach())
    # where the reference's exp(lgamma) form overflows fp32 (alpha > 35) the kernel stays finite
    big = torch.tensor([0.3, 1.0, 60.0, 1.0]).view(1, 4, 1, 1).cuda()
    _, lbig = evidential_head_loss(big, torch.zeros(1, 1, 1, 1).cuda())
    assert torch.isfinite(lbig).all()


@pytest.mark.parametrize("kind", ["dropout2d", "elementwise"])
def test_in_engine_philox_dropout(kind):
    """Dropout multipliers drawn inside the engine (Philox4x32-10 keyed by torch's CUDA generator) instead of torch.bernoulli
    tensors: (a) torch.manual_seed reproduces a run and the generator offset advances, (b) the multipliers are 0 or
    1/(1-p) with the right keep rate, independent across sites / samples / channels, (c) forward AND backward used exactly
    those multipliers: replaying them through the recorded-mask arguments is bit-identical."""
    cfg = O.NetConfig(3, 2, 2, 8)
    st = O.init_state(cfg, 11)
    p = 0.3
    dropout, cf = ((p, p, p), (0.0, 0.0)) if kind == "dropout2d" else ((0.0, 0.0, 0.0), (p, p))
    model = build_model(cfg, st, dropout=dropout, center_final=cf)
    model.train()
    net = model.model
    assert net.engine_rng
    g = torch.Generator().manual_seed(3)
    N = 6
    x = torch.rand(N, 2, 3, 64, 64, generator=g).cuda()
    y = torch.rand(N, 2, 1, 64, 64, generator=g).cuda()

    def run(xin):
        model.zero_grad()
        xg = xin.clone().requires_grad_(True)
        p1, p2 = model(xg)
        model.loss_fn.forward(p1, p2, y, reduce_mean=False).mean().backward()
        return p1.detach().clone(), xg.grad.clone(), net.flat_gradients().clone()

    torch.manual_seed(123)
    off0 = torch.cuda.default_generators[0].get_offset()
    a = run(x)
    assert torch.cuda.default_generators[0].get_offset() == off0 + 4
    plan = next(iter(net._plans.values()))
    ndc = plan.num_double_convs
    sites = range(ndc) if kind == "dropout2d" else range(ndc, ndc + 1 + cfg.num_subnetworks)
    masks = {s: plan.dropout_mask(s) for s in sites}
    b = run(x)                      # next offset: different masks
    assert not torch.equal(a[0], b[0])
    torch.manual_seed(123)
    c = run(x)                      # same seed, same offset: identical run
    assert all(torch.equal(u, v) for u, v in zip(a, c))
    # (b) statistics
    allm = torch.cat([m.flatten() for m in masks.values()])
    vals = torch.unique(allm)
    assert len(vals) == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1 / (1 - p)) < 1e-6
    keep = float((allm > 0).float().mean())
    sigma = (p * (1 - p) / allm.numel()) ** 0.5
    assert abs(keep - (1 - p)) < 5 * sigma, (keep, allm.numel())
    ms = list(masks.values())
    assert not torch.equal(ms[0][0], ms[0][1]) and (len(ms) < 2 or ms[0].shape != ms[1].shape or not torch.equal(ms[0], ms[1]))
    # (c) replay through the recorded-mask path
    if kind == "dropout2d":
        net.mask_override = {s: m for s, m in masks.items()}
    else:
        net.elem_mask_override = {"center": masks[ndc], **{f"final{s}": masks[ndc + 1 + s] for s in range(cfg.num_subnetworks)}}
    d = run(x)
    assert all(torch.equal(u, v) for u, v in zip(a, d))
    report(f"in-engine Philox {kind}: keep rate {keep:.4f} (1 - p = {1 - p}), {allm.numel()} multipliers, replay bit-identical")


def test_two_forwards_then_the_first_ones_backward():
    """torch's autograd keeps every live graph's saved tensors; the engine keeps one set of saved activations per plan,
    so a second forward of the same geometry while the first graph is still alive runs on a second plan — the first
    graph's backward then gives exactly the gradient of an undisturbed run (the reference behaviour), and the extra
    plan is reused, not leaked, afterwards."""
    fx = load_npz("mini_s2_step.npz")
    cfg = cfg_from_meta(fx["meta"])
    model = build_model(cfg, state_from(fx, "init/"))
    model.eval()  # no running-statistics side effects between the calls
    g = torch.Generator().manual_seed(8)
    xa = torch.rand(3, 2, 2, 32, 32, generator=g).cuda()
    xb = torch.rand(3, 2, 2, 32, 32, generator=g).cuda()
    ref = []
    for x in (xa, xb):
        model.zero_grad()
        p1, p2 = model(x)
        (p1.square().mean() + p2.mean()).backward()
        ref.append(model.model.flat_gradients().clone())
    n_plans = len(model.model._plans)
    model.zero_grad()
    a1, a2 = model(xa)
    b1, b2 = model(xb)          # same geometry, first graph still alive
    (a1.square().mean() + a2.mean()).backward()
    assert torch.equal(model.model.flat_gradients(), ref[0])
    model.zero_grad()
    (b1.square().mean() + b2.mean()).backward()
    assert torch.equal(model.model.flat_gradients(), ref[1])
    assert len(model.model._plans) == n_plans + 1
    del a1, a2, b1, b2
    model.zero_grad()
    p1, p2 = model(xa)
    (p1.square().mean() + p2.mean()).backward()
    assert torch.equal(model.model.flat_gradients(), ref[0]) and len(model.model._plans) == n_plans + 1


@pytest.mark.gpu
def test_deepcopy_and_pickle_of_a_model_that_has_run():
    """`copy.deepcopy(model)` (EMA / SWA callbacks) and `torch.save(model)` after the engine has built its plans and moved the
    parameters into its flat storage: the copy takes the module tree, parameters and buffers (MimoUNet.__getstate__ leaves the
    plans' ctypes handles, the flat views and reducer hooks behind), rebuilds the rest at its first forward, and computes the
    same bits as the original; the original keeps training."""
    import copy
    import io
    fx = load_npz("mini_s2_step.npz")
    cfg = cfg_from_meta(fx["meta"])
    model = build_model(cfg, state_from(fx, "init/"))
    model.train()
    image, label, perms = (torch.from_numpy(fx[f"s0/{k}"]).cuda() for k in ("image", "label", "perms"))
    opt = model.configure_optimizers()["optimizer"]
    model.training_step_with_perms(image, label, None, perms)["loss"].backward()
    opt.step()
    twin = copy.deepcopy(model)
    buf = io.BytesIO()
    torch.save(model, buf)
    buf.seek(0)
    loaded = torch.load(buf, weights_only=False)
    outs = [m.training_step_with_perms(image, label, None, perms) for m in (model, twin, loaded)]
    for o in outs[1:]:
        assert torch.equal(o["preds"], outs[0]["preds"]) and torch.equal(o["loss"], outs[0]["loss"])
    for m in (twin, loaded):  # independent storage: a step of the original does not move the copies
        before = m.model.flat_parameters().clone()
        outs[0]["loss"].backward()
        opt.step()
        outs[0] = model.training_step_with_perms(image, label, None, perms)
        assert torch.equal(before, m.model.flat_parameters())
    bn = lambda m: int(m.model.encoder.in_convs[0].double_conv[1].num_batches_tracked)
    assert bn(model) == 4 and bn(twin) == 2 and bn(loaded) == 2


def test_fp16_range_guard_moves_out_of_range_batchnorm_scales_to_the_fp32_range_mode(monkeypatch):
    """VERDICT r5 missing 5 / item 4d, the coarse remedy: a checkpoint whose BatchNorm scale drives an
    activation past 65 520 — where the reference's fp32 convolution (components.py:23,26) stays finite — is moved to the
    precision mode with fp32 exponent range when it is LOADED (|gamma| * 256 + |beta| >= 65504), and trains / evaluates
    within 1e-3 of the fp32 oracle; a scale that grows past the range during training is caught by the epoch-end check,
    which reports the poisoned step and lets the run continue in the fp32-range mode."""
    fx = load_npz("mini_s2_step.npz")
    cfg = cfg_from_meta(fx["meta"])
    st = state_from(fx, "init/")
    st["core.down2.conv.double_conv.1.weight"] = st["core.down2.conv.double_conv.1.weight"].clone()
    st["core.down2.conv.double_conv.1.weight"][0] = 3.0e4  # activations of that channel reach ~1e5
    model = build_model(cfg, st)  # build_model: load_state_dict, then set_precision("split16") — the guard runs again below
    assert model.model._geom.precision == "split16"
    model.load_state_dict({"model." + k: v for k, v in st.items()})
    assert model.model._geom.precision == "fp32" and model.model.precision_guard[0] == "split16"
    image, label, perms = (torch.from_numpy(fx[f"s0/{k}"]) for k in ("image", "label", "perms"))
    model.train()
    out = model.training_step_with_perms(image.cuda(), label.cuda(), None, perms.cuda())
    out["loss"].backward()
    ts = O.TrainState(cfg=cfg, st={k: v.clone() for k, v in st.items()}, loss_buffer=O.LossBuffer(cfg.num_subnetworks, 0.3, 10))
    ref = O.train_step(ts, image, label, None, perms, apply_optimizer=False)
    half = cfg.out_channels // 2
    preds = out["preds"].view(image.shape[0], cfg.num_subnetworks, half, *image.shape[-2:]).cpu()
    e_out = rel_err(preds, ref["out"][:, :, :half])
    e_loss = abs(out["loss"].item() - float(ref["total"])) / abs(float(ref["total"]))
    report(f"fp16 range guard: train out {e_out:.2e} loss {e_loss:.2e} in mode {model.model._geom.precision}")
    assert torch.isfinite(out["loss"]) and e_out < TOL and e_loss < TOL and model.model.numerics_status() == 0
    model.on_train_epoch_end()  # nothing to report
    # a scale that leaves the range DURING training: the step that overflowed is reported once, the run continues in fp32
    m2 = build_model(cfg, state_from(fx, "init/"))
    m2.train()
    with torch.no_grad():
        dict(m2.model.named_parameters())["core.down2.conv.double_conv.1.weight"][0] = 3.0e4
    m2.training_step_with_perms(image.cuda(), label.cuda(), None, perms.cuda())
    with pytest.raises(FloatingPointError, match="continues in 'fp32'"):
        m2.on_train_epoch_end()
    assert m2.model._geom.precision == "fp32"
    out2 = m2.training_step_with_perms(image.cuda(), label.cuda(), None, perms.cuda())
    assert torch.isfinite(out2["loss"]) and m2.model.numerics_status() == 0
    m2.on_train_epoch_end()
    # switched off: the mode stays (and the overflow is only reported)
    monkeypatch.setenv("MIMO_FP16_RANGE_GUARD", "0")
    m3 = build_model(cfg, st)
    m3.load_state_dict({"model." + k: v for k, v in st.items()})
    assert m3.model._geom.precision == "split16"


def test_numerics_status_reports_an_fp16_range_overflow_instead_of_silent_nans(monkeypatch):
    """The default split16 forward carries fp16 (hi, lo) pairs: an ACTIVATION >= 65520 (here: a BatchNorm scale of 3e5)
    turns into fp16 inf and poisons the next convolution's output where the reference's fp32 path does not.  (Weights are
    range-safe since round 5: test_weights_beyond_the_old_fixed_fp16_scale_stay_finite_and_close.)  The kernels record
    non-finite BatchNorm statistics / logits in the plan's status word; `check_numerics` (called by the Lightning epoch-end
    hooks) turns it into an error that says so, the fp32 mode runs the same parameters cleanly, and the word is cleared
    by the read."""
    monkeypatch.setenv("MIMO_FP16_RANGE_GUARD", "0")  # this test is about the REPORT; the guard has its own test above
    fx = load_npz("mini_s2_step.npz")
    cfg = cfg_from_meta(fx["meta"])
    model = build_model(cfg, state_from(fx, "init/"))
    x = torch.rand(3, 2, 32, 32, generator=torch.Generator().manual_seed(3)).cuda()
    lab = torch.rand(3, 1, 32, 32, generator=torch.Generator().manual_seed(4)).cuda()
    model.train()
    out = model.training_step({"image": x, "label": lab}, 0)
    assert torch.isfinite(out["loss"]) and model.model.numerics_status() == 0
    model.on_train_epoch_end()  # nothing recorded: no error
    w = dict(model.model.named_parameters())["core.down2.conv.double_conv.1.weight"]
    with torch.no_grad():
        w[0] = 3.0e5  # BatchNorm scale of the block's first convolution: activations up to ~1e6, beyond fp16
    out = model.training_step({"image": x, "label": lab}, 0)
    # the loss may well stay finite: training-mode BatchNorm turns the poisoned channel into NaNs and the ReLU's fmaxf
    # drops them — a silently dead channel; the status word is what tells
    assert model.model.numerics_status(clear=False) & 1
    with pytest.raises(FloatingPointError, match="fp16"):
        model.on_train_epoch_end()
    assert model.model.numerics_status() == 0  # cleared by the check
    model.eval()
    x5 = x[:, None].repeat(1, 2, 1, 1, 1)
    with torch.no_grad():
        model(x5)  # inference plan: BatchNorm + ReLU folded into the convolution epilogue, which checks its input
    assert model.model.numerics_status() & 1
    model(x5)      # eval forward with autograd: the separate BatchNorm + ReLU pass checks
    assert model.model.numerics_status() & 1
    # what a plan recorded survives its eviction from the LRU plan cache (a NaN on a ragged last batch was
    # lost when other geometries pushed its plan out before the epoch-end check)
    model.train()
    model.training_step({"image": x[:2], "label": lab[:2]}, 0)  # records on the batch-2 plan
    net = model.model
    for n in range(3, 3 + net._plan_cache_size + 1):  # push it out
        with torch.no_grad():
            model.eval()
            model(torch.rand(1, 2, 2, 16 * n, 32, device="cuda"))
    assert not any(k[0] == 2 for k in net._plans) and net._evicted_status & 1
    assert net.numerics_status() & 1 and net.numerics_status() == 0 and net._evicted_status == 0
    ref = build_model(cfg, state_from(fx, "init/"), precision="fp32")  # (a fresh loss buffer: the first model's holds a NaN)
    with torch.no_grad():
        dict(ref.model.named_parameters())["core.down2.conv.double_conv.1.weight"][0] = 3.0e5
    ref.train()
    out = ref.training_step({"image": x, "label": lab}, 0)
    assert torch.isfinite(out["loss"]) and ref.model.numerics_status() == 0
