"""Training dynamics beyond the three golden steps: the same small network (S = 2, fbc = 8, 64 x 64, batch 8) trained for 300
Adam steps from the same initial state on the same stream of learnable batches, three ways — the HIP path in split16
(default), the HIP path in fp32 mode, and the CPU oracle (the restatement pinned to the reference).  Rounding differences
are amplified chaotically after some tens of steps, so the curves are compared as curves: printed side by side, with the
mean loss over the last 50 steps.  Diagnostic for the GPU box, not a test:
    python tests/tools/convergence_probe.py [steps=300]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mimo_oracle as O  # noqa: E402
from tests.test_network_gpu import build_model  # noqa: E402
from bench import learnable_label  # noqa: E402


def batches(steps, N, Ci, H, W):
    g = torch.Generator().manual_seed(5)
    for _ in range(steps):
        image = torch.rand(N, Ci, H, W, generator=g)
        label = learnable_label(image, generator=g)
        perms = O.draw_perms(N, 2, generator=g)
        yield image, label, perms


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    cfg = O.NetConfig(2, 2, 2, 8)
    st = O.init_state(cfg, 11)
    N, H, W = 8, 64, 64
    curves = {}
    for mode in ("split16", "fp32"):
        m = build_model(cfg, st, lr=1e-3, precision=mode)
        m.train()
        opt = m.configure_optimizers()["optimizer"]
        c = []
        for image, label, perms in batches(steps, N, 2, H, W):
            opt.zero_grad()
            out = m.training_step_with_perms(image.cuda(), label.cuda(), None, perms.cuda())
            out["loss"].backward()
            opt.step()
            c.append(float(out["loss"]))
        curves["hip " + mode] = c
    ts = O.TrainState(cfg=cfg, st={k: v.clone() for k, v in st.items()}, lr=1e-3, loss_buffer=O.LossBuffer(2, 0.3, 10))
    c = []
    for image, label, perms in batches(steps, N, 2, H, W):
        c.append(float(O.train_step(ts, image, label, None, perms)["total"]))
    curves["oracle (cpu)"] = c
    names = list(curves)
    print("step  " + "  ".join(f"{n:>14s}" for n in names))
    for i in [0, 1, 2, 5, 10, 20, 50, 100, 150, 200, 250, steps - 1]:
        if i < steps:
            print(f"{i:4d}  " + "  ".join(f"{curves[n][i]:14.6f}" for n in names))
    print("mean of the last 50 steps: " + "  ".join(f"{n}: {sum(curves[n][-50:]) / 50:.5f}" for n in names))


if __name__ == "__main__":
    main()
