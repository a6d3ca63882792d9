"""Two builds of libmimo_hip.so in one process (the default one and MIMO_AB_LIB): outputs of the per-operator entry points on
the thin cfg3 layers compared bit for bit (z, BatchNorm statistics, data gradient)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, ".")
from mimo_unet_amd import _lib as L  # noqa: E402

A = L.load()
B = C.CDLL(os.environ["MIMO_AB_LIB"])
for name, (res, args) in L._SIGNATURES.items():
    fn = getattr(B, name)
    fn.restype, fn.argtypes = res, args
st = L.current_stream()


def pad8(c):
    return (c + 7) // 8 * 8


for (N, H, W, Ci, Co) in [(32, 256, 256, 30, 30), (32, 256, 256, 45, 30), (32, 256, 256, 90, 45), (32, 128, 128, 30, 60), (5, 100, 70, 30, 30)]:
    cip, cop = pad8(Ci), pad8(Co)
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(N, H, W, cip, device="cuda", generator=g)
    x[..., Ci:] = 0
    w = torch.randn(Co, Ci, 3, 3, device="cuda", generator=g) / (3.0 * Ci ** 0.5)
    b = torch.randn(Co, device="cuda", generator=g)
    dz = torch.randn(N, H, W, cop, device="cuda", generator=g)
    dz[..., Co:] = 0
    res = []
    for lib in (A, B):
        z = torch.zeros(N, H, W, cop, device="cuda")
        dx = torch.zeros(N, H, W, cip, device="cuda")
        stats = torch.zeros(2, Co, dtype=torch.float64, device="cuda")
        L.check(lib.mimo_op_conv3x3_forward(x.data_ptr(), w.data_ptr(), b.data_ptr(), z.data_ptr(), stats.data_ptr(), N, H, W, Ci, cip, Co, cop, 1, st))
        L.check(lib.mimo_op_conv3x3_dgrad(dz.data_ptr(), w.data_ptr(), dx.data_ptr(), N, H, W, Ci, cip, Co, cop, 1, st))
        torch.cuda.synchronize()
        res.append((z, stats, dx))
    (za, sa, da), (zb, sb, db) = res
    nz = int((za != zb).sum())
    print(f"{Ci}->{Co}@{H}x{W} N={N}: z differs in {nz} of {za.numel()} elements (max |d| {float((za - zb).abs().max()):.3e}); "
          f"stats equal {bool(torch.equal(sa, sb))} (max rel {float(((sa - sb).abs() / sb.abs().clamp_min(1e-30)).max()):.2e}); "
          f"dgrad differs in {int((da != db).sum())} elements (max |d| {float((da - db).abs().max()):.3e})")
    if nz:
        idx = (za != zb).nonzero()[:8].tolist()
        print("   first differing (n, y, x, c):", idx)
