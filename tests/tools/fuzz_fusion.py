"""Random geometries through the public model surface, two fusions of round 4 against their unfused forms:
  * the activation elision (MIMO_FUSE_BN_IN=1, default — BatchNorm + ReLU applied by the readers' loaders) against every
    activation materialised (=0): the arithmetic is the same, so predictions, loss and every gradient must be BIT-identical;
  * the gradients of pooled tensors / of the head input formed by the BatchNorm backward (MIMO_FUSE_BWD_SRC=1, default)
    against pool_bwd / head_bwd as separate kernels (=0): the same per-element values, per-channel sums grouped
    differently — forward quantities and buffers bit-identical, every gradient within 1e-4 of its scale (printed: worst).
Diagnostic for the GPU box, not a test:
    python tests/tools/fuzz_fusion.py [cases=20] [seed=0]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mimo.models.mimo_unet import MimoUnetModel  # noqa: E402


def run(flag, args, state, image, label, perms, rng_state, bwd_src="1"):
    os.environ["MIMO_FUSE_BN_IN"] = flag  # read per plan
    os.environ["MIMO_FUSE_BWD_SRC"] = bwd_src
    torch.manual_seed(0)
    m = MimoUnetModel(**args).cuda().train()
    m.load_state_dict(state)
    m.loss_buffer.get_weights = lambda: torch.ones(args["num_subnetworks"])
    torch.cuda.set_rng_state(rng_state)  # same Dropout2d draws
    out = m.training_step_with_perms(image, label, None, perms)
    out["loss"].backward()
    named = {n: p.grad.clone() for n, p in m.model.named_parameters()}
    return out["loss"].detach().clone(), out["preds"].clone(), m.model.flat_gradients().clone(), m.state_dict(), named


def worst_grad_deviation(a, b):
    """max over parameters of max|a - b| / max|b| (pre-BatchNorm conv biases — zero gradients — on their weight's scale)"""
    worst = 0.0
    for n, y in b.items():
        pre_bn_bias = "double_conv" in n and n.endswith((".0.bias", ".3.bias"))
        scale = b[n[:-4] + "weight"] if pre_bn_bias else y
        worst = max(worst, float((a[n] - y).abs().max()) / (float(scale.abs().max()) + 1e-30))
    return worst


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for i in range(cases):
        S = rng.choice([1, 2, 3])
        f = rng.choice([6, 8, 12, 16, 21, 30, 34])
        N = rng.choice([1, 2, 3, 5, 8])
        H = rng.choice([48, 64, 70, 96, 100, 128, 130, 160, 256])
        W = rng.choice([48, 56, 64, 72, 96, 110, 128, 200, 256])
        Ci = rng.choice([1, 2, 3])
        drop = rng.choice([0.0, 0.0, 0.0, 0.1])
        args = dict(in_channels=Ci, out_channels=2, num_subnetworks=S, filter_base_count=f, center_dropout_rate=0.0,
                    final_dropout_rate=0.0, encoder_dropout_rate=drop, core_dropout_rate=drop, decoder_dropout_rate=drop,
                    loss="laplace_nll", weight_decay=0.0, learning_rate=1e-3, seed=1, loss_buffer_size=10,
                    loss_buffer_temperature=0.3)
        torch.manual_seed(100 + i)
        ref = MimoUnetModel(**args)
        g = torch.Generator().manual_seed(i)
        state = {k: (v + 0.3 * torch.randn(v.shape, generator=g) if ("double_conv.1." in k or "double_conv.4." in k) and
                     (k.endswith("weight") or k.endswith("bias")) else v) for k, v in ref.state_dict().items()}
        image = torch.rand(N, Ci, H, W, device="cuda")
        label = torch.rand(N, 1, H, W, device="cuda")
        perms = torch.stack([torch.randperm(N) for _ in range(S)]).cuda()
        st = torch.cuda.get_rng_state()
        a = run("1", args, state, image, label, perms, st)
        b = run("0", args, state, image, label, perms, st)
        same = all(torch.equal(x, y) for x, y in zip(a[:3], b[:3])) and all(torch.equal(a[3][k], b[3][k]) for k in a[3])
        c = run("1", args, state, image, label, perms, st, bwd_src="0")
        fwd_same = torch.equal(a[0], c[0]) and torch.equal(a[1], c[1]) and all(torch.equal(a[3][k], c[3][k]) for k in a[3])
        dev = worst_grad_deviation(a[4], c[4])
        ok = same and fwd_same and dev <= 1e-4
        bad += not ok
        print(f"S={S} f={f:2d} N={N} {H}x{W} Ci={Ci} drop={drop}: elision {'bit-identical' if same else 'MISMATCH   <-- CHECK'}; "
              f"gradient sources: forward {'bit-identical' if fwd_same else 'MISMATCH   <-- CHECK'}, worst gradient deviation {dev:.1e}"
              f"{'' if dev <= 1e-4 else '   <-- CHECK'}", flush=True)
    os.environ.pop("MIMO_FUSE_BN_IN", None)
    os.environ.pop("MIMO_FUSE_BWD_SRC", None)
    print("mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
