"""What the pieces of the host -> device hand-over cost on this box (DevicePrefetcher, mimo_unet_amd/data.py)."""
import time

import torch


def t(f, n=20):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return (time.perf_counter() - t0) / n * 1e3


torch.cuda.init()
for name, shape in (("cfg3 batch 32 image", (32, 2, 256, 256)), ("cfg3 batch 4 image", (4, 2, 256, 256))):
    page = torch.rand(shape)
    pin = torch.rand(shape).pin_memory()
    dev = torch.empty(shape, device="cuda")
    print(f"{name} ({page.numel() * 4 / 1e6:.1f} MB):")
    print(f"  Tensor.is_pinned() on a pageable tensor   {t(page.is_pinned):8.3f} ms")
    print(f"  Tensor.is_pinned() on a pinned tensor     {t(pin.is_pinned):8.3f} ms")
    print(f"  pageable -> pinned host copy (main thread){t(lambda: pin.copy_(page)):8.3f} ms")

    def up(src, nb):
        dev.copy_(src, non_blocking=nb)
        torch.cuda.synchronize()
    print(f"  pinned -> device, non_blocking + sync     {t(lambda: up(pin, True)):8.3f} ms")
    print(f"  pageable -> device (torch's blocking copy){t(lambda: up(page, False)):8.3f} ms")
