"""Conditioning of the training step along a trajectory (GPU box, diagnostic — not a test).

Trains the HIP path on bench.py's synthetic batch (uniform-random labels, which no image feature predicts), and at
checkpoints evaluates the SAME parameters four ways: HIP split16, HIP fp32, CPU oracle fp32, CPU oracle fp64.
Prints loss and gradient distance to the fp64 oracle for each — the fp32-oracle column is the noise floor any fp32
implementation sits on at that state.

    python tests/tools/diag_trained_state.py [steps=80] [every=20] [batch=8] [data=random|learnable] [first_check=every]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import mimo_oracle as O  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    every = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    data = sys.argv[4] if len(sys.argv) > 4 else "random"
    first = int(sys.argv[5]) if len(sys.argv) > 5 else every
    c = bench.CONFIGS["cfg3"]
    cfg = O.NetConfig(in_channels=c["Ci"], out_channels=c["Co"], num_subnetworks=c["S"], filter_base_count=c["f"])
    probe = bench.make_model(c).cuda().train()
    torch.manual_seed(1)  # same stream of subnetwork permutations as bench.py from here on
    model = bench.make_model(c).cuda().train()
    opt = model.configure_optimizers()["optimizer"]
    g = torch.Generator(device="cuda").manual_seed(100)
    image = torch.rand(B, c["Ci"], c["H"], c["W"], generator=g, device="cuda")
    label = torch.rand(B, 1, c["H"], c["W"], generator=g, device="cuda")
    if data == "learnable":
        label = bench.learnable_label(image)
    batch = {"image": image, "label": label}
    lb_w = torch.ones(c["S"])
    perms = torch.arange(B).repeat(c["S"], 1)
    for i in range(steps + 1):
        if i % every == 0 and i >= first:
            st = {k[len("model."):]: v.detach().cpu().clone() for k, v in model.state_dict().items() if k.startswith("model.")}
            res = {}
            for prec in ("split16", "fp32"):
                probe.load_state_dict(model.state_dict())
                probe.model.set_precision(prec)
                probe.model.mark_parameters_changed()
                probe.loss_buffer.get_weights = lambda: lb_w
                probe.zero_grad()
                out = probe.training_step_with_perms(image, label, None, perms.cuda())
                out["loss"].backward()
                res["hip " + prec] = (float(out["loss"]), {k[len("model."):]: p.grad.detach().cpu().double()
                                                          for k, p in probe.named_parameters() if p.grad is not None})
            for name, dt in (("oracle fp32", torch.float32), ("oracle fp64", torch.float64)):
                ts = O.TrainState(cfg=cfg, st={k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in st.items()},
                                  loss_kind="laplace_nll", loss_buffer=O.LossBuffer(cfg.num_subnetworks, 0.3, 10))
                ts.loss_buffer.get_weights = lambda: lb_w.to(dt)
                ref = O.train_step(ts, image.cpu().to(dt), label.cpu().to(dt), None, perms, apply_optimizer=False)
                res[name] = (float(ref["total"]), {k: v.double() for k, v in ref["grads"].items()})
            l64, g64 = res["oracle fp64"]
            keys = [k for k in g64 if not (k.endswith((".0.bias", ".3.bias")) and "double_conv" in k)]
            n64 = sum(float((g64[k] ** 2).sum()) for k in keys) ** 0.5
            print(f"step {i}: fp64 loss {l64:.6f} |grad| {n64:.3e}", flush=True)
            for name in ("oracle fp32", "hip fp32", "hip split16"):
                l, gr = res[name]
                d = sum(float(((gr[k] - g64[k]) ** 2).sum()) for k in keys) ** 0.5
                print(f"    {name:12s} loss err {abs(l - l64) / abs(l64):.2e}  grad rel-L2 vs fp64 {d / n64:.2e}", flush=True)
        opt.zero_grad()
        out = model.training_step(batch, i)
        out["loss"].backward()
        opt.step()
        if i >= first - 2:
            print(f"  train step {i}: loss {float(out['loss'].detach()):.5f}", flush=True)


if __name__ == "__main__":
    main()
