"""sha256 of (loss, predictions, every gradient, BatchNorm buffers) of two cfg3-shaped training steps at a given batch /
size — run under two builds of the library (MIMO_HIP_LIB) to check that a kernel change is bit-identical."""
import hashlib
import sys

import torch

sys.path.insert(0, ".")
from bench import CONFIGS, make_model  # noqa: E402

N, H = int(sys.argv[1]), int(sys.argv[2])
c = dict(CONFIGS["cfg3"], H=H, W=H)
torch.manual_seed(1)
m = make_model(c).cuda().train()
g = torch.Generator(device="cuda").manual_seed(3)
batch = {"image": torch.rand(N, 2, H, H, device="cuda", generator=g), "label": torch.rand(N, 1, H, H, device="cuda", generator=g)}
h = hashlib.sha256()
for i in range(2):
    m.zero_grad()
    out = m.training_step(batch, i)
    out["loss"].backward()
    for t in (out["loss"].detach(), out["preds"], m.model.flat_gradients(), m.model._flat_buffers):
        h.update(t.detach().float().cpu().numpy().tobytes())
print(f"N={N} {H}x{H}: {h.hexdigest()[:24]} loss {float(out['loss']):.6f}")
