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


# the staging copy inside a loop that does other things between copies (torch's intra-op thread pool asleep when it is called)
import ctypes
page, pin = torch.rand(4, 2, 256, 256), torch.rand(4, 2, 256, 256).pin_memory()
for name, f in (("dst.copy_(src), default threads", lambda: pin.copy_(page)),
                ("ctypes.memmove", lambda: ctypes.memmove(pin.data_ptr(), page.data_ptr(), page.numel() * 4))):
    ts = []
    for _ in range(10):
        time.sleep(0.005)
        t0 = time.perf_counter()
        f()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"2.1 MB staging copy after 5 ms of other work, {name}: median {sorted(ts)[5]:.3f} ms, max {max(ts):.3f} ms ({torch.get_num_threads()} torch threads)")
