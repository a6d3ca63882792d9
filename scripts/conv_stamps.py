"""In-kernel shader-clock stamps of the wave-specialised convolution (a -DMIMO_CONV_STAMPS build of conv_bf16x3.hip
linked as mimo_unet_amd/libmimo_hip_stamps.so): how long one consumer wave / one producer wave of every workgroup waits
at the phase barriers against its total time, summed over the forward and data-gradient launches of a few cfg3 steps.

    MIMO_HIP_LIB=$PWD/mimo_unet_amd/libmimo_hip_stamps.so MIMO_WGRAD_STREAM=0 python scripts/conv_stamps.py
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mimo_unet_amd import _lib as L  # noqa: E402
from mimo_unet_amd.optim import FlatAdam  # noqa: E402

c = dict(bench.CONFIGS["cfg3"]) if hasattr(bench, "CONFIGS") else None
torch.manual_seed(1)
model = bench.make_model(c).cuda().train()
opt = FlatAdam(model.model, lr=1e-3)
g = torch.Generator(device="cuda").manual_seed(100)
image = torch.rand(c["batch"], c["Ci"], c["H"], c["W"], device="cuda", generator=g)
batch = {"image": image, "label": bench.learnable_label(image, generator=g)}
lib = ctypes.CDLL(L.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()


def step(i):
    opt.zero_grad()
    out = model.training_step(batch, i)
    out["loss"].backward()
    opt.step()


for i in range(3):
    step(i)
torch.cuda.synchronize()
assert lib.mimo_debug_conv_stamps(buf) == 0  # clear
n = 3
for i in range(n):
    step(i)
torch.cuda.synchronize()
assert lib.mimo_debug_conv_stamps(buf) == 0
v = [int(x) for x in buf]
for name, o in (("forward", 0), ("data gradient", 4)):
    cw, ct, pw, pt = v[o:o + 4]
    print(f"{name:14s} consumer wave: {100.0 * cw / max(ct, 1):5.1f} % of its time at the phase barriers; "
          f"producer wave: {100.0 * pw / max(pt, 1):5.1f} % waiting (DMA landed + barrier)   [ticks {cw} / {ct}, {pw} / {pt}]")
