"""The host -> device data path (`mimo_unet_amd.data.DevicePrefetcher`, SURVEY §8f "later" row): a training loop fed from
HOST batches of the reference loaders' shape (``mimo/datasets/nyuv2.py:38-53``: ``{"image": [N,C,H,W], "label":
[N,1,H,W]}`` float32) must (1) compute exactly what the loop over resident tensors computes and (2) keep the host
running ahead of the GPU — no blocking call in a step."""
import time

import pytest
import torch

from tests.helpers import report

pytestmark = pytest.mark.gpu


def _model(f=8, S=2, Ci=2):
    from mimo.models.mimo_unet import MimoUnetModel
    torch.manual_seed(0)
    m = MimoUnetModel(in_channels=Ci, out_channels=2, num_subnetworks=S, filter_base_count=f, center_dropout_rate=0.0,
                      final_dropout_rate=0.0, encoder_dropout_rate=0.0, core_dropout_rate=0.0, decoder_dropout_rate=0.0,
                      loss="laplace_nll", weight_decay=0.0, learning_rate=1e-3, seed=0, loss_buffer_size=10,
                      loss_buffer_temperature=0.3).cuda()
    m.train()
    return m, m.configure_optimizers()["optimizer"]


def _host_batches(n, N, Ci, H, W, pinned, with_mask=False, seed=5):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        b = {"image": torch.rand(N, Ci, H, W, generator=g), "label": torch.rand(N, 1, H, W, generator=g)}
        if with_mask:
            b["mask"] = (torch.rand(N, 1, H, W, generator=g) > 0.3).float()
        out.append({k: v.pin_memory() for k, v in b.items()} if pinned else b)
    return out


def _train(batches_on_device, seed=11):
    m, opt = _model()
    torch.manual_seed(seed)  # the permutation streams (CPU + CUDA generators)
    losses = []
    for i, b in enumerate(batches_on_device):
        opt.zero_grad()
        out = m.training_step(b, i)
        out["loss"].backward()
        opt.step()
        losses.append(out["loss"].detach().clone())
    torch.cuda.synchronize()
    return torch.stack(losses).cpu(), m.model.flat_parameters().clone().cpu()


@pytest.mark.parametrize("pinned", [False, True], ids=["pageable", "pinned"])
@pytest.mark.parametrize("depth", [1, 2, 3])
def test_prefetched_loop_is_bit_identical_to_the_resident_loop(pinned, depth):
    """7 distinct batches (more than the ring holds, so every slot is reused) with a mask, then a ragged last batch."""
    from mimo_unet_amd.data import DevicePrefetcher
    host = _host_batches(7, 4, 2, 32, 48, pinned, with_mask=True) + _host_batches(1, 3, 2, 32, 48, pinned, with_mask=True, seed=6)
    ref_l, ref_p = _train([{k: v.cuda() for k, v in b.items()} for b in host])
    pf = DevicePrefetcher(host, device="cuda", depth=depth)
    assert len(pf) == 8
    got_l, got_p = _train(pf)
    assert torch.equal(ref_l, got_l) and torch.equal(ref_p, got_p)
    assert pf.pageable_uploads == (0 if pinned else 3 * 8)


def test_two_epochs_back_to_back_do_not_overwrite_batches_that_pending_steps_still_read():
    """the slots used to belong to one `iter()`; at the end of an epoch they were dropped with up to
    `depth` enqueued steps still waiting to read them, and the next epoch's worker could upload its first batches into the
    same memory at once.  Here the GPU is held back (a long idle kernel in front of the last steps of every epoch) while
    the host walks straight into the next epoch — no synchronisation anywhere in the loop — and the result must still be
    the resident loop's, bit for bit; the same after a loop that left its epoch early."""
    import itertools
    from mimo_unet_amd.data import DevicePrefetcher
    host = _host_batches(5, 4, 2, 32, 48, pinned=True, with_mask=True)
    stall = _sleep_cycles_for(60.0)

    def stalled(batches):  # hold the GPU back in front of the last two steps of the epoch
        for i, b in enumerate(batches):
            if i >= 3:
                torch.cuda._sleep(stall)
            yield b

    resident = [{k: v.cuda() for k, v in b.items()} for b in host]
    ref_l, ref_p = _train(itertools.chain(stalled(resident), stalled(resident), resident))
    for depth in (1, 2):
        pf = DevicePrefetcher(host, device="cuda", depth=depth)
        got_l, got_p = _train(itertools.chain(stalled(pf), stalled(pf), pf))
        assert torch.equal(ref_l, got_l) and torch.equal(ref_p, got_p), f"depth {depth}"
    # an epoch left after two batches (uploads in flight, the steps pending), then a full one
    ref_l, ref_p = _train(itertools.chain(itertools.islice(stalled(resident), 4), resident))
    pf = DevicePrefetcher(host, device="cuda", depth=2)
    got_l, got_p = _train(itertools.chain(itertools.islice(stalled(pf), 4), pf))
    assert torch.equal(ref_l, got_l) and torch.equal(ref_p, got_p)


def test_prefetcher_passes_non_tensors_through_and_rejects_cpu():
    from mimo_unet_amd.data import DevicePrefetcher
    with pytest.raises(ValueError):
        DevicePrefetcher([], device="cpu")
    b = {"image": torch.ones(2, 1, 4, 4), "meta": "patch-17", "label": torch.zeros(2, 1, 4, 4)}
    (out,) = list(DevicePrefetcher([b], device="cuda"))
    assert out["meta"] == "patch-17" and out["image"].is_cuda and float(out["image"].sum()) == 32.0


def test_prefetcher_surfaces_loader_errors_survives_an_early_stop_and_can_be_iterated_again():
    """What a training script does to a loader besides iterating it to the end: an exception raised by the wrapped iterable
    (a corrupt sample) must come out of the consuming loop, not die in the worker thread; a loop that stops early
    (`limit_train_batches`, a `break`) must not leave the next epoch hanging; and a second `iter()` — the next epoch —
    yields the same batches again (every epoch gets its own worker and slots)."""
    from mimo_unet_amd.data import DevicePrefetcher
    host = _host_batches(5, 2, 2, 16, 16, pinned=True)

    class Loader:
        def __init__(self, fail_at=None):
            self.fail_at = fail_at

        def __len__(self):
            return len(host)

        def __iter__(self):
            for i, b in enumerate(host):
                if i == self.fail_at:
                    raise RuntimeError("corrupt sample 3")
                yield b

    pf = DevicePrefetcher(Loader(fail_at=3), device="cuda", depth=2)
    seen = []
    with pytest.raises(RuntimeError, match="corrupt sample 3"):
        for b in pf:
            seen.append(b["image"].clone())
    # everything in front of the failure was delivered, in order
    assert len(seen) == 3 and all(torch.equal(g.cpu(), h["image"]) for g, h in zip(seen, host))

    pf = DevicePrefetcher(Loader(), device="cuda", depth=2)
    for i, b in enumerate(pf):
        if i == 1:
            break  # early stop with uploads in flight
    for epoch in range(2):  # ... and two full epochs afterwards
        got = [b["label"].clone() for b in pf]
        assert len(got) == 5 and all(torch.equal(g.cpu(), h["label"]) for g, h in zip(got, host))


def _sleep_cycles_for(ms):
    """cycles argument of torch.cuda._sleep for about `ms` of device time (calibrated on this box)"""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(1_000_000)
    torch.cuda.synchronize()
    e0.record()
    torch.cuda._sleep(20_000_000)
    e1.record()
    torch.cuda.synchronize()
    return int(20_000_000 * ms / max(e0.elapsed_time(e1), 1e-3))


@pytest.mark.parametrize("pinned", [False, True], ids=["pageable", "pinned"])
def test_steps_fed_from_host_memory_contain_no_blocking_call(pinned):
    """Deterministic form of "the host runs ahead": the stream is held by a ~0.4 s spin kernel, then training steps fed
    through the prefetcher are enqueued.  Any blocking call inside (a pageable copy, an .item(), a synchronise) would
    return only after the spin kernel — the gate event behind it would then be complete when the host gets there.
    The prefetcher's WORKER waits for the GPU before it overwrites a device slot (host-side hand-off, data.py), so the
    consumer runs `depth` = 2 steps ahead on what was uploaded before and would then wait for data: two steps, from
    pinned and from pageable memory alike; with the stream-wait hand-off (MIMO_PREFETCH_HANDOFF=gpu) and pinned batches
    nothing ever waits for the GPU: three steps, more than the ring holds."""
    from mimo_unet_amd.data import DevicePrefetcher
    m, opt = _model(f=30)
    host = _host_batches(8, 4, 2, 256, 256, pinned)
    pf = DevicePrefetcher(host, device="cuda", depth=2)
    it = iter(pf)

    def step(i):
        opt.zero_grad()
        m.training_step(next(it), i)["loss"].backward()
        opt.step()

    for i in range(4):  # plans, pinned slots and device slots exist after these
        step(i)
    torch.cuda.synchronize()
    cycles = _sleep_cycles_for(400.0)
    blocked = 0
    gate = torch.cuda.Event()
    torch.cuda._sleep(cycles)
    gate.record()
    t0 = time.perf_counter()
    nsteps = 3 if pinned and pf.handoff == "gpu" else 2
    for i in range(nsteps):
        step(i)
        blocked += int(gate.query())
    host_s = time.perf_counter() - t0
    torch.cuda.synchronize()
    report(f"{nsteps} steps fed from {'pinned' if pinned else 'pageable'} host batches behind a 0.4 s spin kernel: host returned "
           f"after {host_s * 1e3:.1f} ms, steps that found the gate complete: {blocked}, waits for the worker {pf.starved}")
    assert blocked == 0


def test_host_fed_loop_runs_ahead_like_the_resident_loop():
    """VERDICT r4 item 5 on cfg3 at batch 16, 10 steps: (1) fed from PINNED host tensors (what the reference's loaders yield,
    `pin_memory=True`) through a ring deep enough for the loop (depth 8: the consumer may be that many steps ahead of the GPU,
    200 MB of HBM) the host enqueues the steps in less than half the time the GPU needs — the run-ahead test's criterion;
    (2) at the default depth 2, from pinned and from PAGEABLE tensors, the consumer is held to two batches ahead — it then
    waits for DATA, with the GPU busy: the loop must take no longer than the loop over resident tensors (+ 5 %; + 15 % from pageable memory)."""
    from mimo_unet_amd.data import DevicePrefetcher
    import itertools
    m, opt = _model(f=30)

    def run(source):
        it = iter(source)

        def step(i):
            opt.zero_grad()
            m.training_step(next(it), i)["loss"].backward()
            opt.step()

        for i in range(4):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(10):
            step(i)
        host_s = time.perf_counter() - t0
        torch.cuda.synchronize()
        return host_s, time.perf_counter() - t0

    host = _host_batches(2, 16, 2, 256, 256, pinned=False)
    resident = [{k: v.cuda() for k, v in b.items()} for b in host]
    _, t_res0 = run(itertools.cycle(resident))
    pinned = [{k: v.pin_memory() for k, v in b.items()} for b in host]
    # (wall-clock criterion: best of up to three loops — in one of four shuffled orders of the suite this loop's host time was 103 ms
    # instead of the usual 26-31 with nothing else different; a blocking call in the path would show in all of them, and the gate test
    # above counts blocking calls without a clock)
    h_deep, t_deep, starved_deep = None, None, []
    for _ in range(3):
        pf_deep = DevicePrefetcher(itertools.cycle(pinned), device="cuda", depth=8)
        h, t = run(pf_deep)
        starved_deep.append(pf_deep.starved)
        if h_deep is None or h < h_deep:
            h_deep, t_deep = h, t
        if h_deep < 0.5 * t_deep:
            break
    h_pin, t_pin = run(DevicePrefetcher(itertools.cycle(pinned), device="cuda", depth=2))
    h_page, t_page = run(DevicePrefetcher(itertools.cycle(host), device="cuda", depth=2))
    _, t_res1 = run(itertools.cycle(resident))
    # (the reference is measured in front of and behind the fed loops: 130 ms loops see the clock state the tests before them
    # left, and this test failed once in a shuffled run of the suite with a single reference in front)
    t_res = max(t_res0, t_res1)
    report(f"cfg3 batch 16, 10 steps: resident {t_res0 * 1e3:.1f} / {t_res1 * 1e3:.1f} ms (before / after); pinned host batches, depth 8: "
           f"{t_deep * 1e3:.1f} ms (host enqueued in {h_deep * 1e3:.1f} ms, waits for the worker per loop {starved_deep}); depth 2: pinned {t_pin * 1e3:.1f} ms (host {h_pin * 1e3:.1f} ms), "
           f"pageable {t_page * 1e3:.1f} ms (host {h_page * 1e3:.1f} ms) - host times at depth 2 include waiting for data")
    assert h_deep < 0.5 * t_deep, (h_deep, t_deep)
    assert max(t_deep, t_pin) < 1.05 * t_res, (t_res0, t_res1, t_deep, t_pin)
    # (pageable batches - not what the reference's loaders yield - go through the driver's staged copy, whose speed is the
    # host's: 1.01 x and 1.08 x the resident loop on two boxes of the same pool)
    assert t_page < 1.15 * t_res, (t_res0, t_res1, t_page)
