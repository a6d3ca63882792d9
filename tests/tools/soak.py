"""Long run of the headline configuration as a training loop would drive it: cfg3 (2 -> 1 ch, 256 x 256, S = 2, fbc = 30,
batch 32), `steps` Adam steps fed from a pool of pinned host batches through `DevicePrefetcher`, `on_train_epoch_end`
(the status word of the split16 path) every 250 steps.  Prints the loss every 100 steps, the sustained rate per 250-step
block (clock / thermal drift shows here), device memory after the first block and at the end, and the prefetcher's
counters.  Diagnostic for the GPU box, not a test:
    python tests/tools/soak.py [steps=1500]"""
import itertools
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import CONFIGS, learnable_label, make_model  # noqa: E402
from mimo_unet_amd.data import DevicePrefetcher  # noqa: E402


def used_mb():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    c = CONFIGS["cfg3"]
    torch.manual_seed(1)
    model = make_model(c).cuda()
    model.train()
    opt = model.configure_optimizers()["optimizer"]
    g = torch.Generator().manual_seed(7)
    pool = []
    for _ in range(16):  # 16 different batches, cycled (drawing 4 M uniform numbers per step on the host would be the bottleneck)
        image = torch.rand(c["batch"], c["Ci"], c["H"], c["W"], generator=g)
        pool.append({"image": image.pin_memory(), "label": learnable_label(image, generator=g).pin_memory()})
    pf = DevicePrefetcher(itertools.islice(itertools.cycle(pool), steps), device="cuda", depth=2)
    losses, block_t0, mem_first = [], None, None
    print(f"# cfg3 batch {c['batch']}, {steps} steps, precision {os.environ.get('MIMO_PRECISION', 'split16')}, "
          f"16 host batches cycled through DevicePrefetcher(depth=2)")
    print("#  step        loss   images/s of the last 250 steps   device memory in use (MB, hipMemGetInfo: plan + torch allocator)")
    t_start = time.perf_counter()
    for i, batch in enumerate(pf):
        if i % 250 == 0:
            torch.cuda.synchronize()
            now = time.perf_counter()
            rate = 250 * c["batch"] / (now - block_t0) if block_t0 is not None else float("nan")
            block_t0 = now
            if i:
                model.on_train_epoch_end()  # raises on an fp16 overflow / non-finite statistics recorded in the status word
                if mem_first is None:
                    mem_first = used_mb()
        opt.zero_grad()
        out = model.training_step(batch, i)
        out["loss"].backward()
        opt.step()
        if i % 100 == 0 or i == steps - 1:
            loss = float(out["loss"])  # (a host sync every 100 steps, as a logger would do)
            losses.append((i, loss))
            extra = f"   {rate:10.1f}   {used_mb():10.0f}" if i % 250 == 0 and i else ""
            print(f"{i:7d}  {loss:10.5f}{extra}", flush=True)
    torch.cuda.synchronize()
    total = time.perf_counter() - t_start
    model.on_train_epoch_end()
    mem_last = used_mb()
    finite = all(l == l and abs(l) < 1e6 for _, l in losses)
    print(f"# {steps} steps in {total:.1f} s = {steps * c['batch'] / total:.1f} images/s incl. the first (plan-building) steps; "
          f"loss {losses[0][1]:.4f} -> {losses[-1][1]:.4f}, all finite: {finite}; device memory in use after 250 steps "
          f"{(mem_first or mem_last):.0f} MB, at the end {mem_last:.0f} MB; prefetcher: waited for its worker "
          f"{pf.starved} times, {pf.pageable_uploads} pageable uploads")
    if not finite or (mem_first and mem_last > mem_first * 1.01):
        raise SystemExit("soak: non-finite loss or growing memory")


if __name__ == "__main__":
    main()
