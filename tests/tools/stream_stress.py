"""Side stream against one stream at the bench geometry: cfg3 (S = 2, fbc = 30, 256 x 256) at 4 and at 32 images, `steps` Adam
steps from the same state with MIMO_WGRAD_STREAM=0 and =1; the flat gradient buffer after every backward and the parameters at
the end must agree bit for bit (same kernels, only the stream changes).  The small batch is where the main stream runs furthest
ahead of the side stream — the timing in which round 5's max |dz| release race showed on a small test geometry.
Diagnostic for the GPU box, not a test:
    python tests/tools/stream_stress.py [steps=40]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import CONFIGS, learnable_label, make_model  # noqa: E402


def run(mode, batch, steps):
    os.environ["MIMO_WGRAD_STREAM"] = mode  # read when the plan is created
    c = dict(CONFIGS["cfg3"], batch=batch)
    torch.manual_seed(1)
    model = make_model(c).cuda()
    model.train()
    opt = model.configure_optimizers()["optimizer"]
    g = torch.Generator(device="cuda").manual_seed(100)
    sums = []
    for i in range(steps):
        image = torch.rand(batch, c["Ci"], c["H"], c["W"], device="cuda", generator=g)
        label = learnable_label(image, generator=g)
        opt.zero_grad()
        model.training_step({"image": image, "label": label}, i)["loss"].backward()
        sums.append(model.model.flat_gradients().clone())
        opt.step()
    torch.cuda.synchronize()
    return sums, [p.detach().clone() for p in model.parameters()]


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    bad = 0
    for batch in (4, 32):
        g0, p0 = run("0", batch, steps)
        for rep in range(2):
            g1, p1 = run("1", batch, steps)
            diff = [i for i, (a, b) in enumerate(zip(g0, g1)) if not torch.equal(a, b)]
            pd = sum(int(not torch.equal(a, b)) for a, b in zip(p0, p1))
            print(f"cfg3 batch {batch}, {steps} steps, side stream run {rep}: gradient buffers that differ from the one-stream run: "
                  f"{len(diff)} {diff[:5]}; parameter tensors that differ at the end: {pd}", flush=True)
            bad += len(diff) + pd
    if bad:
        raise SystemExit("stream_stress: the side stream changed results")
    print("stream_stress: bit-identical")


if __name__ == "__main__":
    main()
