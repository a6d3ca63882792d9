"""Data-parallel path on the GPU box: two ranks (sharing the single GPU, gloo transport) run the real
training step on their shard; after the overlapped all-reduce both hold the mean of the two shard
gradients and take the same fused-Adam step."""
import os

import pytest
import torch

from tests.helpers import free_port, cfg_from_meta, load_npz, state_from

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from mimo_unet_amd.ddp import FlatGradientAllReducer, shard_batch
    from tests.test_network_gpu import build_model
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    fx = load_npz("mini_s2_step.npz")
    cfg = cfg_from_meta(fx["meta"])
    g = torch.Generator().manual_seed(3)
    full = {"image": torch.rand(4, 2, 32, 32, generator=g), "label": torch.rand(4, 1, 32, 32, generator=g)}
    shard = {k: v.cuda() for k, v in shard_batch(full, rank, world).items()}
    perms = torch.stack([torch.arange(2), torch.tensor([1, 0])]).cuda()
    model = build_model(cfg, state_from(fx, "init/"))
    model.train()
    opt = model.configure_optimizers()["optimizer"]
    red = FlatGradientAllReducer(bucket_bytes=1 << 18)
    red.attach(model.model)
    opt.reduce_scale = red.scale
    calls = []
    inner = model.model.grad_ready_hook
    model.model.grad_ready_hook = lambda flat, b, e: (calls.append((b, e)), inner(flat, b, e))
    local = {}
    # local (un-reduced) gradient of this shard, for the reference mean
    m2 = build_model(cfg, state_from(fx, "init/"))
    m2.train()
    m2.training_step_with_perms(shard["image"], shard["label"], None, perms)["loss"].backward()
    local = m2.model.flat_gradients().clone().cpu()
    opt.zero_grad()
    model.training_step_with_perms(shard["image"], shard["label"], None, perms)["loss"].backward()
    red.finish()
    reduced = model.model.flat_gradients().clone().cpu()
    opt.step()
    torch.cuda.synchronize()
    # numpy (pickled by value): torch tensors on an mp.Queue travel as shared fds, which break when
    # the sender exits before the parent has received them
    q.put((rank, local.numpy(), reduced.numpy(), model.model.flat_parameters().clone().cpu().numpy(), calls))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_step_gloo_on_gpu():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    res = [(r, torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(c), d) for r, a, b, c, d in res]
    (_, l0, r0, p0, c0), (_, l1, r1, p1, c1) = res
    assert torch.allclose(r0, l0 + l1, rtol=1e-6, atol=1e-9) and torch.equal(r0, r1)  # sum on every rank
    assert torch.equal(p0, p1)                                                       # identical Adam step (scale 1/2)
    # one announcement per backward stage, walking the flat buffer from its tail to its head without gaps
    assert len(c0) == 8 and c0[0][1] == r0.numel() and c0[-1][0] == 0
    assert all(c0[i][0] == c0[i + 1][1] for i in range(7))


def _accum_worker(rank, world, port, q):
    """Two micro-batches per optimiser step: the first under no_sync(), the second announces the accumulated sum."""
    import torch.distributed as dist
    from mimo_unet_amd.ddp import FlatGradientAllReducer
    from tests.test_network_gpu import build_model
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    fx = load_npz("mini_s2_step.npz")
    cfg = cfg_from_meta(fx["meta"])
    g = torch.Generator().manual_seed(11 + rank)
    micro = [(torch.rand(2, 2, 32, 32, generator=g).cuda(), torch.rand(2, 1, 32, 32, generator=g).cuda()) for _ in range(2)]
    perms = torch.stack([torch.arange(2), torch.tensor([1, 0])]).cuda()
    ones = lambda: torch.ones(cfg.num_subnetworks, device="cuda")
    local = []
    for im, lb in micro:  # local gradients of each micro-batch, no reducer
        m = build_model(cfg, state_from(fx, "init/"))
        m.train()
        m.loss_buffer.get_weights = ones
        m.training_step_with_perms(im, lb, None, perms)["loss"].backward()
        local.append(m.model.flat_gradients().clone().cpu())
    model = build_model(cfg, state_from(fx, "init/"))
    model.train()
    model.loss_buffer.get_weights = ones
    red = FlatGradientAllReducer(bucket_bytes=1 << 18)
    red.attach(model.model)
    with red.no_sync():
        model.training_step_with_perms(*micro[0], None, perms)["loss"].backward()
    assert not red.busy and not red.issued
    model.training_step_with_perms(*micro[1], None, perms)["loss"].backward()
    red.finish()
    reduced = model.model.flat_gradients().clone().cpu()
    # the mistake the guard catches: a third backward into already-reduced gradients
    refused = False
    try:
        model.training_step_with_perms(*micro[0], None, perms)["loss"].backward()
    except RuntimeError as e:
        refused = "no_sync" in str(e)
    torch.cuda.synchronize()
    q.put((rank, (local[0] + local[1]).numpy(), reduced.numpy(), refused))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_accumulation_under_the_reducer_sums_each_micro_batch_once():
    """with the reducer attached, micro-batch 1 used to be all-reduced when its ranges became final and again
    together with micro-batch 2.  no_sync() defers the exchange to the last micro-batch; a backward into gradients that
    were already reduced is refused."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_accum_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, s0, r0, f0), (_, s1, r1, f1) = [(r, torch.from_numpy(a), torch.from_numpy(b), c) for r, a, b, c in res]
    assert torch.equal(r0, r1) and torch.allclose(r0, s0 + s1, rtol=1e-5, atol=1e-8)
    assert f0 and f1


# cfg3's 60.3 MB of gradients under FlatGradientAllReducer's defaults (merge until >= 4 MB are pending): heads + decoders + up3
# + up2 (6.9 MB) | up1 (20.7) | down4 (16.6) | down3 (12.4) | down2 + encoders (3.6, flushed by finish())
N_COLLECTIVES_CFG3 = 5


def _run_bench(extra_env, *argv):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **extra_env)
    env.pop("MIMO_PARITY_LOG", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], cwd=root, env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_spawns_its_own_ranks_gloo_sharing_the_gpu():
    """`python bench.py --gpus 2` with no torch.distributed environment starts two rank processes itself (the form
    the driver uses); here over gloo with both ranks on GPU 0 so that it runs on a single-GPU box.  Strong scaling:
    the global batch of 4 is sharded 2 + 2; rank-0 parameters are broadcast at start and all ranks end bit-identical."""
    line = _run_bench({"MIMO_BENCH_BACKEND": "gloo"}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
                      "--scaling", "strong", "--profile-steps", "0", "--no-cpu-baseline")
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == 2
    cfg = line["config"]
    assert cfg["global_batch"] == 4 and cfg["per_gpu_batch"] == 2
    assert cfg["world_size"] == 2 and cfg["backend"] == "gloo" and cfg["rccl_ranks"] is None  # RCCL did not run
    assert cfg["params_bit_identical_across_ranks"] is True
    assert line["value"] > 0 and "roofline" not in line
    # both regimes in the one line: the timed one (strong: 4 global) and the other from a second timed pass
    assert cfg["strong_images_per_s"] == line["value"] and cfg["weak_images_per_s"] > 0
    assert cfg["weak_per_gpu_batch"] == 4 and cfg["strong_global_batch"] == 4


def test_bench_eight_ranks_gloo_sharing_the_gpu():
    """The 8-GPU regime of BASELINE config 3 as a FUNCTIONAL run (world 2 was the only size ever
    exercised): `python bench.py --gpus 8 --batch 32` starts eight ranks, here over gloo on GPU 0 — global batch 32 sharded
    4 per rank (ddp.shard_batch), the backward of every rank issues the same five bucketed collectives, and all eight ranks
    end with bit-identical parameters.  Not a performance configuration."""
    line = _run_bench({"MIMO_BENCH_BACKEND": "gloo", "OMP_NUM_THREADS": "2"}, "--gpus", "8", "--steps", "2", "--warmup", "1",
                      "--batch", "32", "--scaling", "strong", "--profile-steps", "0", "--no-cpu-baseline", "--no-strict",
                      "--one-regime")
    cfg = line["config"]
    assert line["n_gpus"] == 8 and cfg["world_size"] == 8 and cfg["backend"] == "gloo" and cfg["rccl_ranks"] is None
    assert cfg["global_batch"] == 32 and cfg["per_gpu_batch"] == 4
    # cfg3: 15.06 M parameters = 60.3 MB of gradients, out as five buckets (N_COLLECTIVES_CFG3 above)
    assert cfg["collectives_per_step"] == N_COLLECTIVES_CFG3, (cfg["collectives_per_step"], cfg["collective_mbytes"])
    assert abs(sum(cfg["collective_mbytes"]) - 60.3) < 0.5, cfg["collective_mbytes"]
    assert cfg["params_bit_identical_across_ranks"] is True
    assert len(cfg["rank_devices"]) == 8 and line["value"] > 0


@pytest.mark.parametrize("algo", ["reduce_scatter"])
def test_bench_two_ranks_reduce_scatter_all_gather(algo):
    """The reduce-scatter + all-gather form of the exchange (ddp.FlatGradientAllReducer(algorithm=...), MIMO_DDP_ALGO) through
    the real step: two ranks over gloo on GPU 0, parameters bit-identical across the ranks at the end."""
    line = _run_bench({"MIMO_BENCH_BACKEND": "gloo", "MIMO_DDP_ALGO": algo}, "--gpus", "2", "--steps", "2", "--warmup", "1",
                      "--batch", "4", "--profile-steps", "0", "--no-cpu-baseline", "--no-strict", "--one-regime")
    cfg = line["config"]
    assert cfg["ddp_algorithm"] == algo and cfg["collectives_per_step"] == N_COLLECTIVES_CFG3, cfg["collective_mbytes"]
    assert cfg["params_bit_identical_across_ranks"] is True


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
def test_bench_two_ranks_rccl():
    line = _run_bench({}, "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "4", "--profile-steps", "0",
                      "--no-cpu-baseline")
    # default regime = strong: the batch of 4 is the global batch, 2 per GPU; the weak regime rides along
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["global_batch"] == 4
    assert line["config"]["weak_images_per_s"] > 0 and line["config"]["weak_per_gpu_batch"] == 4
    assert line["config"]["rccl_ranks"] == 2 and line["config"]["backend"] == "nccl" and line["config"]["collectives_per_step"] == N_COLLECTIVES_CFG3
    assert line["config"]["params_bit_identical_across_ranks"] is True
    assert len(set(line["config"]["rank_devices"])) == 2  # one GPU per rank
