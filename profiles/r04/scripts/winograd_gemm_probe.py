#!/usr/bin/env python3
"""Measured bound for VERDICT r3 item 1 (Winograd F(2x2,3x3) on 480->480 @ 32x32, batch 32, split16): how long does the GEMM
part alone take when a tuned library runs it?  16 transform positions x [8192 tiles x 480] x [480 x 480]; split16 = three
16-bit products per fp32 product, here as ONE batched GEMM with K = 3 x 480 (a_hi|a_hi|a_lo) . (w_hi|w_lo|w_hi)^T — the
library accumulates in fp32, exactly the three-MFMA scheme.  hipBLASLt through torch.bmm; operands and result resident in
HBM, transforms NOT included (input transform: read 63 MB fp32, write 16 x 8192 x 1440 x 2 B = 377 MB; output transform:
read the fp32 products 252 MB, write 63 MB).  Prints one line per variant; no product code involved."""
import torch

dev = "cuda"
T, C, K3 = 32 * 16 * 16, 480, 3 * 480


def bench(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


g = torch.Generator(device=dev).manual_seed(1)
for dt in (torch.float16, torch.bfloat16):
    a3 = torch.randn(16, T, K3, device=dev, generator=g).to(dt)
    w3 = torch.randn(16, K3, C, device=dev, generator=g).to(dt)
    a1, w1 = a3[:, :, :C].contiguous(), w3[:, :C, :].contiguous()
    us3 = bench(lambda: torch.bmm(a3, w3))
    us1 = bench(lambda: torch.bmm(a1, w1))
    fl = 2.0 * 16 * T * C * C
    print(f"{dt}: 16 x [{T} x {K3}] x [{K3} x {C}] (split16, K = 3 x 480): {us3:7.1f} us = {3 * fl / us3 / 1e6:6.1f} TFLOP/s of 16-bit MFMA work;"
          f"   one product (K = 480): {us1:7.1f} us = {fl / us1 / 1e6:6.1f} TFLOP/s")
    try:
        us32 = bench(lambda: torch.bmm(a3, w3, out_dtype=torch.float32))
        print(f"{dt}: same with an fp32 result (what the output transform needs): {us32:7.1f} us")
    except Exception as e:  # noqa: BLE001
        print(f"{dt}: fp32 result not available through torch.bmm here ({type(e).__name__})")
print("direct convolution, same layer (conv3x3_wide_kernel, profiles/r04/final/conv_layers.txt): 320.8 us forward")
