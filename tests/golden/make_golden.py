"""Generate golden vectors from the REAL reference (``/root/reference``).

Run in the build container only (the reference does not exist on the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

It imports ``mimo.models.mimo_components.{model,loss_buffer}``, ``mimo.losses``
and ``mimo.models.utils`` from the reference tree, drives them exactly as the
reference's own training loop does (notebook cell 14 /
``mimo/models/mimo_unet.py:115-144``; Lightning zeroes grads every step), and
stores inputs + outputs as ``.npz`` fixtures next to this file.  Nothing of the
reference's source is copied: the fixtures are data only.

Random draws that the HIP path cannot reproduce bit-for-bit (torch's Philox
stream) are made explicit: permutations are recorded, and dropout masks are
drawn here and fed to the reference by swapping ``F.dropout2d`` /``F.dropout``
for an ``x * mask`` with a recorded mask (the arithmetic stays the reference's).
"""
import os
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != "/root/repo"]
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from mimo.losses import EvidentialLoss, GaussianNLL, LaplaceNLL  # noqa: E402
from mimo.models.mimo_components.loss_buffer import LossBuffer  # noqa: E402
from mimo.models.mimo_components.model import MimoUNet  # noqa: E402
from mimo.models.utils import apply_input_transform, compute_uncertainties, repeat_subnetworks  # noqa: E402

assert os.path.abspath(sys.modules["mimo"].__file__).startswith(REF)
torch.set_num_threads(4)
torch.use_deterministic_algorithms(True)


def npd(t):
    return t.detach().cpu().numpy().copy()


def recorded_perms(seed, n, S):
    """Replay of the draws inside apply_input_transform (utils.py:27-36) for
    irp=0, reps=1: randperm(N) then S x randperm(N)."""
    torch.manual_seed(seed)
    main = torch.randperm(n)
    return torch.stack([main[torch.randperm(n)] for _ in range(S)])


def train_fixture(name, *, Ci, Co, S, f, N, H, W, use_mask, steps, seed, T=0.3, lr=1e-3, wd=0.0,
                  loss_kind="laplace_nll"):
    torch.manual_seed(seed)
    net = MimoUNet(in_channels=Ci, out_channels=Co, num_subnetworks=S, filter_base_count=f)
    net.train()
    crit = LaplaceNLL() if loss_kind == "laplace_nll" else GaussianNLL()
    lb = LossBuffer(subnetworks=S, temperature=T, buffer_size=10)
    opt = torch.optim.Adam(net.parameters(), lr=lr, weight_decay=wd)
    g = torch.Generator().manual_seed(seed + 1)
    fx = {"meta": np.array([Ci, Co, S, f, N, H, W, int(use_mask), steps]), "lr": np.float64(lr), "wd": np.float64(wd),
          "temperature": np.float64(T), "loss_kind": np.array(loss_kind), "torch_version": np.array(torch.__version__)}
    for k, v in net.state_dict().items():
        fx["init/" + k] = npd(v)
    for it in range(steps):
        image = torch.rand(N, Ci, H, W, generator=g)
        label = torch.rand(N, Co // 2, H, W, generator=g)
        mask = (torch.rand(N, 1, H, W, generator=g) > 0.25).float() if use_mask else None
        pseed = 1000 + 17 * it + seed
        perms = recorded_perms(pseed, N, S)
        torch.manual_seed(pseed)
        xt, yt, mt = apply_input_transform(image, label, mask, num_subnetworks=S)
        assert torch.equal(xt, torch.stack([image[perms[s]] for s in range(S)], 1))
        xt.requires_grad_(True)
        opt.zero_grad()
        out = net(xt)
        p1, p2 = out[:, :, :Co // 2], out[:, :, Co // 2:]
        raw = crit.forward(p1, p2, yt, reduce_mean=False, mask=mt)
        loss = raw.mean(dim=(0, 2, 3, 4))
        weights = lb.get_weights()
        lb.add(loss.detach())
        total = (loss * weights).mean()
        total.backward()
        fx[f"s{it}/image"], fx[f"s{it}/label"], fx[f"s{it}/perms"] = npd(image), npd(label), npd(perms)
        if use_mask:
            fx[f"s{it}/mask"] = npd(mask)
        fx[f"s{it}/loss"], fx[f"s{it}/weights"], fx[f"s{it}/total"] = npd(loss), npd(weights), npd(total)
        if it == 0:
            fx["s0/out"] = npd(out)
            fx["s0/dx"] = npd(xt.grad)
            for k, p in net.named_parameters():
                fx["s0/grad/" + k] = npd(p.grad)
        opt.step()
        if it == 0:
            for k, v in net.state_dict().items():
                if "running" in k or "num_batches" in k:
                    fx["s0/after/" + k] = npd(v)
    for k, v in net.state_dict().items():
        fx["final/" + k] = npd(v)
    fx["final/loss_buffer"] = npd(lb.buffer)
    np.savez_compressed(os.path.join(HERE, name), **fx)
    print(name, sum(v.nbytes for v in fx.values()) / 1e6, "MB raw")


def odd_size_fixture():
    fx = {}
    for tag, (N, H, W) in {"50x70": (2, 50, 70), "100x100": (1, 100, 100), "128x160": (1, 128, 160)}.items():
        torch.manual_seed(7)
        net = MimoUNet(in_channels=3, out_channels=2, num_subnetworks=2, filter_base_count=4)
        if "init/encoder.in_convs.0.double_conv.0.weight" not in fx:
            for k, v in net.state_dict().items():
                fx["init/" + k] = npd(v)
        x = torch.rand(N, 2, 3, H, W, generator=torch.Generator().manual_seed(H * W))
        net.train()
        out_train = net(x)
        net.eval()  # running stats now hold one update
        with torch.no_grad():
            out_eval = net(x)
        fx[tag + "/x"], fx[tag + "/out_train"], fx[tag + "/out_eval"] = npd(x), npd(out_train), npd(out_eval)
        for k, v in net.state_dict().items():
            if "running" in k:
                fx[tag + "/after/" + k] = npd(v)
    np.savez_compressed(os.path.join(HERE, "odd_sizes.npz"), **fx)
    print("odd_sizes.npz")


def loss_fixture():
    fx = {}
    g = torch.Generator().manual_seed(3)
    ls_vals = torch.tensor([-20.0, -11.6, -11.5, -1.0, 0.0, 3.0, 6.9, 6.95, 9.0])
    mu = torch.randn(4, 9, generator=g)
    mu[0, 0] = 0.5
    y = torch.randn(4, 9, generator=g)
    y[0, 0] = 0.5  # sign(0) case
    ls = ls_vals[None, :].repeat(4, 1)
    mask = (torch.rand(4, 9, generator=g) > 0.3).float()
    for kind, crit in (("laplace", LaplaceNLL()), ("gaussian", GaussianNLL())):
        a = mu.clone().requires_grad_(True)
        b = ls.clone().requires_grad_(True)
        raw = crit.forward(a, b, y, reduce_mean=False, mask=mask)
        raw.sum().backward()
        fx[kind + "/raw"], fx[kind + "/dmu"], fx[kind + "/dls"] = npd(raw), npd(a.grad), npd(b.grad)
        fx[kind + "/mean"] = npd(crit.forward(mu, ls, y))
        fx[kind + "/std"] = npd(crit.std(mu, ls))
        fx[kind + "/dist_param"] = npd(crit.calculate_dist_param(crit.std(mu, ls)))
        fx[kind + "/dist_param_log"] = npd(crit.calculate_dist_param(crit.std(mu, ls), log=True))
        for S in (1, 2, 16):
            p1 = torch.randn(2, S, 1, 5, 6, generator=g)
            p2 = torch.randn(2, S, 1, 5, 6, generator=g)
            m, al, ep = compute_uncertainties(crit, p1, p2)
            fx[f"{kind}/unc{S}/p1"], fx[f"{kind}/unc{S}/p2"] = npd(p1), npd(p2)
            fx[f"{kind}/unc{S}/mean"], fx[f"{kind}/unc{S}/alea"], fx[f"{kind}/unc{S}/epi"] = npd(m), npd(al), npd(ep)
    fx["mu"], fx["y"], fx["ls"], fx["mask"] = npd(mu), npd(y), npd(ls), npd(mask)
    lb = LossBuffer(subnetworks=3, temperature=0.3, buffer_size=10)
    seq = torch.rand(12, 3, generator=g) * 2
    ws = []
    for i in range(12):
        ws.append(lb.get_weights().clone())
        lb.add(seq[i])
    fx["lossbuf/seq"], fx["lossbuf/weights"], fx["lossbuf/final"] = npd(seq), npd(torch.stack(ws)), npd(lb.buffer)
    lb0 = LossBuffer(subnetworks=2, temperature=1.0, buffer_size=0)
    lb0.add(torch.ones(2))
    fx["lossbuf/size0_weights"] = npd(lb0.get_weights())
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **fx)
    print("losses.npz")


def mc_dropout_fixture():
    """Eval-mode BN + active Dropout2d (ensemble.py:54-66), 4 passes, masks recorded."""
    Ci, Co, S, f, N, H, W, passes, p = 3, 2, 1, 6, 2, 32, 32, 4, 0.1
    torch.manual_seed(11)
    net = MimoUNet(in_channels=Ci, out_channels=Co, num_subnetworks=S, filter_base_count=f,
                   encoder_dropout_rate=p, core_dropout_rate=p, decoder_dropout_rate=p)
    g = torch.Generator().manual_seed(12)
    # make the running stats non-trivial: two training passes without dropout noise mattering
    net.train()
    for _ in range(2):
        net(torch.rand(N, S, Ci, H, W, generator=g))
    net.eval()
    for m in net.modules():  # ensemble.py:63-66
        if m.__class__.__name__.startswith("Dropout"):
            m.train()
    fx = {"meta": np.array([Ci, Co, S, f, N, H, W, passes]), "p": np.float64(p)}
    for k, v in net.state_dict().items():
        fx["state/" + k] = npd(v)
    x = torch.rand(N, Ci, H, W, generator=g)
    fx["x"] = npd(x)
    names = {id(m): n for n, m in net.named_modules()}
    rec = []
    orig2d, orig = F.dropout2d, F.dropout

    def fake2d(inp, p=0.5, training=True, inplace=False):
        if not training or p == 0:
            return inp
        mk = (torch.rand(inp.shape[0], inp.shape[1], generator=g) >= p).float() / (1 - p)
        rec.append(mk)
        return inp * mk[:, :, None, None]

    F.dropout2d = fake2d
    try:
        p1s, p2s = [], []
        xr = repeat_subnetworks(x, S)
        for i in range(passes):
            rec.clear()
            with torch.no_grad():
                out = net(xr)
            p1s.append(out[:, :, :Co // 2])
            p2s.append(out[:, :, Co // 2:])
            # call order of Dropout2d modules == forward order of DoubleConvs
            for j, mk in enumerate(rec):
                fx[f"pass{i}/mask{j}"] = npd(mk)
            fx[f"pass{i}/nmask"] = np.array(len(rec))
    finally:
        F.dropout2d, F.dropout = orig2d, orig
    p1, p2 = torch.cat(p1s, 1), torch.cat(p2s, 1)
    mean, al, ep = compute_uncertainties(LaplaceNLL(), p1, p2)
    fx["p1"], fx["p2"], fx["mean"], fx["alea"], fx["epi"] = npd(p1), npd(p2), npd(mean), npd(al), npd(ep)
    np.savez_compressed(os.path.join(HERE, "mc_dropout.npz"), **fx)
    print("mc_dropout.npz")


def elem_dropout_fixture():
    """Element-wise center / final nn.Dropout (model.py:213 and :277-281 of the reference) in a
    training-mode forward + backward; masks recorded in call order (center, final 0..S-1)."""
    Ci, Co, S, f, N, H, W, pc, pf = 2, 2, 2, 4, 3, 32, 32, 0.2, 0.3
    torch.manual_seed(21)
    net = MimoUNet(in_channels=Ci, out_channels=Co, num_subnetworks=S, filter_base_count=f,
                   center_dropout_rate=pc, final_dropout_rate=pf)
    net.train()
    g = torch.Generator().manual_seed(22)
    fx = {"meta": np.array([Ci, Co, S, f, N, H, W]), "pc": np.float64(pc), "pf": np.float64(pf)}
    for k, v in net.state_dict().items():
        fx["init/" + k] = npd(v)
    x = torch.rand(N, S, Ci, H, W, generator=g).requires_grad_(True)
    y = torch.rand(N, S, Co // 2, H, W, generator=g)
    rec = []
    orig = F.dropout

    def fake(inp, p=0.5, training=True, inplace=False):
        if not training or p == 0:
            return inp
        mk = (torch.rand(inp.shape, generator=g) >= p).float() / (1 - p)
        rec.append(mk)
        return inp * mk

    F.dropout = fake
    try:
        out = net(x)
    finally:
        F.dropout = orig
    assert len(rec) == 1 + S
    loss = LaplaceNLL().forward(out[:, :, :Co // 2], out[:, :, Co // 2:], y, reduce_mean=False).mean(dim=(0, 2, 3, 4))
    loss.mean().backward()
    fx["x"], fx["y"], fx["out"], fx["loss"], fx["dx"] = npd(x), npd(y), npd(out), npd(loss), npd(x.grad)
    fx["mask/center"] = npd(rec[0])
    for s in range(S):
        fx[f"mask/final{s}"] = npd(rec[1 + s])
    for k, p in net.named_parameters():
        fx["grad/" + k] = npd(p.grad)
    np.savez_compressed(os.path.join(HERE, "elem_dropout.npz"), **fx)
    print("elem_dropout.npz")


def evidential_fixture():
    """EvidentialUnetModel (evidential_unet.py:74-118) on the real MimoUNet + EvidentialLoss: the class
    itself needs lightning, so its forward glue (unsqueeze, softplus heads) is restated here; backbone
    and loss are the reference's.  One training-mode forward + backward."""
    Ci, f, N, H, W = 3, 4, 2, 32, 32
    torch.manual_seed(31)
    net = MimoUNet(in_channels=Ci, out_channels=4, num_subnetworks=1, filter_base_count=f)
    net.train()
    g = torch.Generator().manual_seed(32)
    fx = {"meta": np.array([Ci, 4, 1, f, N, H, W])}
    for k, v in net.state_dict().items():
        fx["init/" + k] = npd(v)
    x = torch.rand(N, Ci, H, W, generator=g).requires_grad_(True)
    y = torch.rand(N, 1, H, W, generator=g)
    mask = (torch.rand(N, H, W, generator=g) > 0.2).float()
    out = net(torch.unsqueeze(x, dim=1)).squeeze(dim=1)
    mu, logv, logalpha, logbeta = torch.unbind(out, axis=1)
    sp = torch.nn.Softplus()
    ev = torch.stack([mu, sp(logv), sp(logalpha) + 1, sp(logbeta)], dim=1)
    crit = EvidentialLoss(coeff=1.0)
    loss = crit(ev, y, mask=mask)
    loss.mean().backward()
    fx["x"], fx["y"], fx["mask"], fx["ev"], fx["loss"], fx["dx"] = npd(x), npd(y), npd(mask), npd(ev), npd(loss), npd(x.grad)
    fx["aleatoric_var"], fx["epistemic_var"] = npd(crit.aleatoric_var(ev)), npd(crit.epistemic_var(ev))
    for k, p in net.named_parameters():
        fx["grad/" + k] = npd(p.grad)
    # loss extremes on hand-made NIG parameters
    e2 = torch.tensor([[[0.3]], [[1e-3]], [[1.0001]], [[1e-4]]]).reshape(1, 4, 1, 1).repeat(1, 1, 1, 3)
    e2[0, 1, 0, 1], e2[0, 2, 0, 1], e2[0, 3, 0, 1] = 5.0, 9.0, 4.0
    e2[0, 1, 0, 2], e2[0, 2, 0, 2], e2[0, 3, 0, 2] = 0.5, 1.5, 0.7
    y2 = torch.tensor([0.1, -2.0, 0.3]).reshape(1, 1, 1, 3)
    fx["ext/ev"], fx["ext/y"], fx["ext/loss"] = npd(e2), npd(y2), npd(crit(e2, y2))
    np.savez_compressed(os.path.join(HERE, "evidential.npz"), **fx)
    print("evidential.npz")


def amp_fixture(name, src, *, Ci, Co, S, f, N, H, W, use_mask, seed, T=0.3):
    """The reference's PRODUCTION precision (scripts/train/train_ndvi.py:71, train_nyuv2_depth.py:74: Lightning
    precision="16-mixed") and its bf16 twin, from the real reference modules under torch.autocast: Lightning's
    MixedPrecisionPlugin runs training_step (forward AND loss) inside `torch.autocast(device, dtype)` and the backward
    outside it, on scaler.scale(loss).  There is no GPU here, so this is torch's CPU autocast policy (conv2d -> 16-bit,
    reflection_pad2d -> fp32, everything else by type promotion: BatchNorm keeps 16-bit activations with fp32
    statistics); the fp16 run uses a FIXED loss scale (recorded) instead of GradScaler's dynamic one.

    Same seed, parameters and step-0 inputs as the fp32 fixture `src` (asserted), so that each tensor here has its fp32
    counterpart there: the distance between the two is the reference's own precision loss in that mode."""
    ref = np.load(os.path.join(HERE, src))
    torch.manual_seed(seed)
    net = MimoUNet(in_channels=Ci, out_channels=Co, num_subnetworks=S, filter_base_count=f)
    state0 = {k: v.clone() for k, v in net.state_dict().items()}
    for k, v in state0.items():
        assert np.array_equal(npd(v), ref["init/" + k]), k
    g = torch.Generator().manual_seed(seed + 1)
    image = torch.rand(N, Ci, H, W, generator=g)
    label = torch.rand(N, Co // 2, H, W, generator=g)
    mask = (torch.rand(N, 1, H, W, generator=g) > 0.25).float() if use_mask else None
    perms = recorded_perms(1000 + seed, N, S)
    assert np.array_equal(npd(image), ref["s0/image"]) and np.array_equal(npd(perms), ref["s0/perms"])
    fx = {"meta": ref["meta"], "src": np.array(src), "torch_version": np.array(torch.__version__),
          "policy": np.array("torch.autocast('cpu', dtype): forward + loss inside, backward outside (Lightning MixedPrecisionPlugin)")}
    for tag, dtype, scale in (("bf16", torch.bfloat16, 1.0), ("fp16", torch.float16, 1024.0)):
        net.load_state_dict(state0)
        net.train()
        net.zero_grad()
        crit = LaplaceNLL()
        lb = LossBuffer(subnetworks=S, temperature=T, buffer_size=10)
        torch.manual_seed(1000 + seed)
        xt, yt, mt = apply_input_transform(image, label, mask, num_subnetworks=S)
        xt.requires_grad_(True)
        with torch.autocast("cpu", dtype=dtype):
            out = net(xt)
            p1, p2 = out[:, :, :Co // 2], out[:, :, Co // 2:]
            raw = crit.forward(p1, p2, yt, reduce_mean=False, mask=mt)
            loss = raw.mean(dim=(0, 2, 3, 4))
            weights = lb.get_weights()
            lb.add(loss.detach())
            total = (loss * weights).mean()
        assert out.dtype == dtype
        (total * scale).backward()
        fx[f"{tag}/loss_scale"] = np.float64(scale)
        fx[f"{tag}/out"], fx[f"{tag}/out_dtype"] = npd(out.float()), np.array(str(out.dtype))
        fx[f"{tag}/loss"], fx[f"{tag}/total"] = npd(loss.float()), npd(total.float())
        fx[f"{tag}/loss_dtype"] = np.array(str(loss.dtype))
        fx[f"{tag}/dx"] = npd(xt.grad / scale)
        for k, p in net.named_parameters():
            assert p.grad.dtype == torch.float32
            fx[f"{tag}/grad/" + k] = npd(p.grad / scale)
        for k, v in net.state_dict().items():
            if "running" in k:
                fx[f"{tag}/after/" + k] = npd(v)
        # eval-mode forward with the INITIAL running statistics (mean 0, var 1), as the fp32 eval comparisons do
        net.load_state_dict(state0)
        net.eval()
        with torch.no_grad(), torch.autocast("cpu", dtype=dtype):
            fx[f"{tag}/out_eval"] = npd(net(xt.detach()).float())
    np.savez_compressed(os.path.join(HERE, name), **fx)
    print(name, sum(v.nbytes for v in fx.values()) / 1e6, "MB raw")


if __name__ == "__main__":
    # BASELINE config[0]: synthetic 3ch 64x64, S=1, fbc=8, batch 4
    train_fixture("cfg1_step.npz", Ci=3, Co=2, S=1, f=8, N=4, H=64, W=64, use_mask=False, steps=3, seed=1)
    # S=2 mini with mask: concat/stack/loss-buffer weights != 1
    train_fixture("mini_s2_step.npz", Ci=2, Co=2, S=2, f=4, N=3, H=32, W=32, use_mask=True, steps=3, seed=2)
    # gaussian loss, weight decay, two targets (Co=4)
    train_fixture("mini_gauss_step.npz", Ci=3, Co=4, S=2, f=2, N=2, H=32, W=48, use_mask=False, steps=2, seed=3,
                  wd=1e-2, loss_kind="gaussian_nll")
    odd_size_fixture()
    loss_fixture()
    mc_dropout_fixture()
    elem_dropout_fixture()
    evidential_fixture()
    # the reference's production precision (Lightning "16-mixed") and bf16 autocast, one training step each
    amp_fixture("amp_cfg1.npz", "cfg1_step.npz", Ci=3, Co=2, S=1, f=8, N=4, H=64, W=64, use_mask=False, seed=1)
    amp_fixture("amp_mini_s2.npz", "mini_s2_step.npz", Ci=2, Co=2, S=2, f=4, N=3, H=32, W=32, use_mask=True, seed=2)
