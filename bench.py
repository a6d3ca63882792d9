#!/usr/bin/env python3
"""Headline benchmark: MIMO U-Net training throughput (images/s) at 256x256, S=2, fbc=30.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--config cfg3|cfg2|cfg4]

N > 1 without a torch.distributed environment: this process starts the N ranks itself (a child
`python -m torch.distributed.run --nproc-per-node N bench.py ...`, before anything here touches the GPU) and
exits with the child's status; under torch.distributed.run (RANK / WORLD_SIZE set) it is one rank.

One "step" = what Lightning runs per batch for the reference's `MimoUnetModel`
(mimo/models/mimo_unet.py:115-144 + backward + Adam): draw the S batch permutations,
forward (gather fused into the first kernel), Laplace NLL, loss-buffer weighting, backward,
[gradient all-reduce over RCCL when N>1], fused Adam.  Synthetic inputs resident in HBM: image ~ U[0,1),
label ~ U[0,1) (`--labels learnable`: a smoothed channel mix of the image + uniform noise — the U[0,1)
labels make the NLL's scale head collapse on single pixels after ~100 steps of the same batch in every arithmetic, DESIGN 4,
which a default run never reaches), the same batch every
step, PyTorch default random init, fp32 storage (MIMO_PRECISION=bf16-mixed | 16-mixed select the 16-bit storage modes — reduced precision, never the
headline metric).

Scaling: "strong" (default) = the config's batch is the GLOBAL batch, sharded over the ranks (cfg3 = 32
global, 4 per GPU at 8) — `value` at N > 1 is that regime; "weak" = the per-GPU batch is fixed at the config's batch
(32: the reference README's SEN12TP batch size with Lightning-DDP semantics, batch_size = per device).  At N > 1 the
line carries both (`config.strong_images_per_s`, `config.weak_images_per_s`), each from its own timed pass.

`value_strict` / `ms_per_step_strict`: the same timed pass of a second model under MIMO_WGRAD_NP=3 (three bf16-pair MFMAs per
product in the weight gradient instead of the default's two fp16 MFMAs with the activation as one fp16 value): the fp32-class
figure next to the headline (`--no-strict` skips it).

The timed region runs WITHOUT instrumentation.  A second, shorter pass with HIP events recorded on the launch
stream around every kernel class and every resolution tier gives the roofline numbers; at N=1 the CPU oracle
is timed on this box's host cores for the same workload shape.  ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs[2] geometry; batch 32 (README: --batch_size 32)
    "cfg3": dict(name="cfg3: SEN12TP-shape synthetic 2->1 ch, 256x256, S=2, fbc=30, laplace_nll",
                 Ci=2, Co=2, S=2, f=30, H=256, W=256, batch=32),
    # BASELINE.json configs[1]
    "cfg2": dict(name="cfg2: NYUv2-shape synthetic 3->1 ch, 256x256, S=2, fbc=21, laplace_nll",
                 Ci=3, Co=2, S=2, f=21, H=256, W=256, batch=64),
    # BASELINE.json configs[3] geometry (S=4 head-width stress; 128 global = 16 per GPU at 8); run with
    # MIMO_PRECISION=bf16 for its arithmetic
    "cfg4": dict(name="cfg4: synthetic 2->1 ch, 256x256, S=4, fbc=30, laplace_nll",
                 Ci=2, Co=2, S=4, f=30, H=256, W=256, batch=16),
}
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 at the vector rate
# split16 arithmetic: every algorithmic product is three 16-bit MFMAs (hi.hi + hi.lo + lo.hi), so the
# ceiling for ALGORITHMIC flops is the dense bf16/fp16 MFMA peak (~2500 TFLOP/s) divided by 3
SPLIT16_PEAK_TFLOPS = 2500.0 / 3.0
HBM_PEAK_GBS = 8000.0


def learnable_label(image, noise=0.05, generator=None):
    """Synthetic regression target that the image actually predicts: a 5x5-smoothed non-linear mix of the
    input channels plus uniform noise of known width (so a Laplace/Gaussian NLL has a real optimum instead of
    the all-noise labels torch.rand gives, where BatchNorm ends up normalising near-constant channels)."""
    import torch.nn.functional as F
    mix = 0.6 * image[:, :1] + 0.4 * image[:, -1:] ** 2
    smooth = F.avg_pool2d(F.pad(mix, (2, 2, 2, 2), mode="reflect"), 5, stride=1)
    return smooth + noise * (torch.rand(smooth.shape, device=image.device, generator=generator) - 0.5)


def conv_layers(c):
    """(tier, Cin, Cout, H, W, kernel) of every convolution of the network, per SURVEY 8(d) "Formula for any
    config" (reference layer widths: model.py:150-175,190-243,260-297)."""
    S, f, Ci, Co, H, W = c["S"], c["f"], c["Ci"], c["Co"], c["H"], c["W"]
    w = f * S
    L = []

    def dc(tier, cin, mid, cout, times=1):
        h, ww = H >> tier, W >> tier
        L.extend([(tier, cin, mid, h, ww, 3), (tier, mid, cout, h, ww, 3)] * times)

    dc(0, Ci, f, f, S)
    dc(1, f, 2 * f, 2 * f, S)
    dc(2, 2 * w, 4 * w, 4 * w)
    dc(3, 4 * w, 8 * w, 8 * w)
    dc(4, 8 * w, 8 * w, 8 * w)
    dc(3, 16 * w, 8 * w, 4 * w)
    dc(2, 8 * w, 4 * w, 2 * w)
    dc(1, 4 * w, 2 * w, w)
    dc(0, w + f, (w + f) // 2, f, S)
    L.extend([(0, f, Co, H, W, 1)] * S)
    return L


def algorithmic_bytes_per_image(c, elem_bytes=4):
    """train = 3 x forward, forward = sum over conv layers of (input + output) tensor bytes (fp32; 2 bytes
    per element in the 16-bit storage modes), every other operator fused away.  Returns ([bytes per tier], total).
    cfg3: 525.3 / 188.7 / 94.4 / 47.2 / 5.9 MB = 861.5 MB per image; cfg4 in bf16: 1050 MB."""
    tiers = [0.0] * 5
    for t, cin, cout, h, w, _ in conv_layers(c):
        tiers[t] += 3.0 * elem_bytes * (cin + cout) * h * w
    return tiers, sum(tiers)


def make_model(c):
    from mimo.models.mimo_unet import MimoUnetModel
    return MimoUnetModel(in_channels=c["Ci"], out_channels=c["Co"], num_subnetworks=c["S"], filter_base_count=c["f"],
                         center_dropout_rate=0.0, final_dropout_rate=0.0, encoder_dropout_rate=0.0, core_dropout_rate=0.0,
                         decoder_dropout_rate=0.0, loss="laplace_nll", weight_decay=0.0, learning_rate=1e-3, seed=1,
                         loss_buffer_size=10, loss_buffer_temperature=0.3)


def csrc_hash():
    """Content hash of the kernel sources (mimo_unet_amd/csrc/*.hip, *.h) — what ties a committed counter summary to
    the build that runs (no git on the GPU box: hashed from the files themselves)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mimo_unet_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic():
    """HBM bytes from the committed counter passes of this same command (profiles/<round>/final/pmc_traffic.json,
    written from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs — counters cannot be collected from inside the
    timed run).  Only a summary collected on THIS build's kernel sources counts (its "csrc_hash" == csrc_hash()):
    (None, why) otherwise — a stale summary must not be reported as this run's traffic."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "final", "pmc_traffic.json")))
    if not files:
        return None, "no committed pmc_traffic.json"
    try:
        with open(files[-1]) as fh:
            pt = json.load(fh)
    except (ValueError, OSError):
        return None, "unreadable " + os.path.relpath(files[-1], ROOT)
    rel = os.path.relpath(files[-1], ROOT)
    if pt.get("csrc_hash") != csrc_hash():
        return None, f"{rel} was collected on other kernel sources (csrc_hash {pt.get('csrc_hash')} != {csrc_hash()})"
    return pt, rel


def cpu_baseline(c, batch=8, warmup=2, steps=5, threads=None, labels="uniform"):
    """The CPU oracle (restatement of the reference's step, pinned to reference goldens) on the host cores of
    this box: same network shape, SURVEY 8(d)'s bounded sample (batch 8, 2 warm-up + 5 timed steps, ~20 s).
    Thread count: torch's CPU convolutions stop scaling (and collapse when oversubscribed) well below this
    box's core count (measured on the 256-core GPU box: 8 thr 2.6, 16 thr 3.4, 32 thr 3.0, 64 thr 1.7, 128 thr
    0.8 images/s), so the run uses min(cores, MIMO_BENCH_CPU_THREADS or 16) threads and reports that."""
    from oracle import mimo_oracle as O
    threads = threads or min(os.cpu_count() or 1, int(os.environ.get("MIMO_BENCH_CPU_THREADS", "16")))
    torch.set_num_threads(threads)
    cfg = O.NetConfig(c["Ci"], c["Co"], c["S"], c["f"])
    ts = O.TrainState(cfg=cfg, st=O.init_state(cfg, 1), loss_buffer=O.LossBuffer(c["S"], 0.3, 10))
    g = torch.Generator().manual_seed(1)
    image = torch.rand(batch, c["Ci"], c["H"], c["W"], generator=g)
    label = learnable_label(image, generator=g) if labels == "learnable" else torch.rand(batch, 1, c["H"], c["W"], generator=g)
    times = []
    for i in range(warmup + steps):
        perms = O.draw_perms(batch, c["S"], generator=g)
        t0 = time.perf_counter()
        O.train_step(ts, image, label, None, perms)
        times.append(time.perf_counter() - t0)
    dt = sum(times[warmup:]) / steps
    return {"value": round(batch / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(),
            "threads": torch.get_num_threads(), "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"{steps} timed steps ({warmup} warm-up) of the same network shape at batch {batch}, fp32, "
                      f"torch {torch.__version__} CPU, {dt * 1e3:.0f} ms/step"}


def visible_gpu_count(sysfs="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs this process tree will see, WITHOUT any HIP / torch.cuda call (the parent of an N-rank run must not
    initialise the GPU before it starts its children): KFD topology nodes with SIMDs, cut down by
    ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES.  None when sysfs does not say."""
    if not os.path.isdir(os.path.dirname(os.path.dirname(sysfs))):
        return None  # sysfs does not say (masked in a container, or no KFD driver): the ranks' own device check decides
    try:
        n = 0
        for node in sorted(os.listdir(sysfs)):
            with open(os.path.join(sysfs, node, "properties")) as fh:
                props = dict(ln.split()[:2] for ln in fh if len(ln.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
    except (OSError, ValueError):
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def spawn_ranks(args):
    """Parent of an N-rank run: start the ranks as a CHILD process tree; this process never touches the GPU (no
    torch.cuda / HIP call: a process that has initialised HIP must not fork-exec), forwards the children's output
    and returns their status."""
    have = visible_gpu_count()
    if have is not None and have < args.gpus and os.environ.get("MIMO_BENCH_BACKEND", "nccl") == "nccl":
        raise SystemExit(f"--gpus {args.gpus} but only {have} GPU(s) visible (MIMO_BENCH_BACKEND=gloo lets ranks share one)")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="the config's batch (per GPU when weak, global when strong)")
    # strong (default) = the config's batch is the GLOBAL batch (cfg3 = 32 global, 4 per GPU at 8 GPUs);
    # weak = the config's batch per GPU (Lightning-DDP's batch_size semantics).  At one GPU the two are the same run.
    ap.add_argument("--scaling", default=os.environ.get("MIMO_BENCH_SCALING", "strong"), choices=["weak", "strong"])
    ap.add_argument("--labels", default="uniform", choices=["uniform", "learnable"],
                    help="uniform (default): SURVEY 8d's label ~ U[0,1); learnable: label = learnable_label(image)")
    ap.add_argument("--one-regime", action="store_true", help="N > 1: skip the second timed pass in the other scaling regime")
    ap.add_argument("--no-strict", action="store_true",
                    help="skip the second timed pass under MIMO_WGRAD_NP=3 (value_strict: the three-MFMA weight gradient)")
    ap.add_argument("--profile-steps", type=int, default=5, help="steps of the instrumented second pass (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # the data path: every step's batch starts in HOST memory and reaches HBM through
    # mimo_unet_amd.data.DevicePrefetcher (pinned double buffer, copy stream, event hand-off) — a diagnostic regime:
    # `value` of the default run is quoted with the inputs resident in HBM, as the contract says
    ap.add_argument("--host-batches", default=None, choices=["pinned", "pageable"],
                    help="feed each step from host tensors (a DataLoader's output: pinned with pin_memory=True) through DevicePrefetcher")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))

    c = dict(CONFIGS[args.config])
    if args.batch:
        c["batch"] = args.batch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and (world > 1 or "RANK" in os.environ):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the line would not describe the run that was asked for")
    if args.scaling == "strong" and c["batch"] % world:
        raise SystemExit(f"strong scaling: global batch {c['batch']} does not split over {world} ranks")
    B = c["batch"] // world if args.scaling == "strong" else c["batch"]  # per-GPU batch
    # one process per GPU over RCCL ("nccl").  MIMO_BENCH_BACKEND=gloo lets several ranks share one GPU
    # (functional check of the data-parallel path on a single-GPU box; not a performance configuration)
    backend = os.environ.get("MIMO_BENCH_BACKEND", "nccl")
    if world > 1 and backend == "nccl" and torch.cuda.device_count() < world:
        # (the spawning parent could not tell from sysfs, or the ranks were started by hand)
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} GPU(s) visible "
                         "(MIMO_BENCH_BACKEND=gloo lets ranks share one)")
    dev = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev)
    dist = None
    # MIMO_BENCH_FORCE_DIST=1: take the data-parallel code path (RCCL init, bucketed all-reduce overlapped with
    # the encoder backward, barrier, max-over-ranks) even with one rank — a functional check on a 1-GPU box
    force_dist = world == 1 and os.environ.get("MIMO_BENCH_FORCE_DIST", "0") != "0" and "RANK" in os.environ
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
        if dist.get_backend() != backend:
            raise SystemExit(f"process group runs on {dist.get_backend()!r}, {backend!r} was requested")
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the {backend} process group has {dist.get_world_size()} rank(s)")
        # one GPU per rank under RCCL (ranks may share a device only in the gloo functional check)
        devs = [None] * dist.get_world_size()
        # identity of the physical device: host + PCI address (+ uuid where torch exposes it) — not the device index,
        # which is 0 on every rank when the launcher masks one GPU per rank
        p = torch.cuda.get_device_properties(dev)
        known = hasattr(p, "pci_bus_id") or hasattr(p, "uuid")  # else nothing tells two masked devices apart: no check
        pci = ":".join(f"{getattr(p, a, -1):x}" for a in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
        dist.all_gather_object(devs, f"{socket.gethostname()}:{pci}:{getattr(p, 'uuid', dev)}" if known
                               else f"{socket.gethostname()}:rank{rank}")
        if backend == "nccl" and len(set(devs)) != len(devs):
            raise SystemExit(f"RCCL ranks share a GPU: {devs}")

    from mimo_unet_amd.ddp import FlatGradientAllReducer
    precision = os.environ.get("MIMO_PRECISION", "split16")

    class Run:
        """model + optimiser (+ reducer, + loss scaler) of one timed pass; the plans read the environment when they are
        created (first step), so a second Run under MIMO_WGRAD_NP=3 gives the strict-arithmetic figure in the same process"""

        def __init__(self):
            torch.manual_seed(1)
            self.model = make_model(c).cuda()
            self.model.train()
            if dist is not None:
                # DDP semantics: every rank starts from rank 0's parameters and buffers (not from a shared seed)
                for t in list(self.model.parameters()) + list(self.model.buffers()):
                    dist.broadcast(t.data, 0)
            self.opt = self.model.configure_optimizers()["optimizer"]
            self.opt.reduce_scale = 1.0 / world
            self.reducer = FlatGradientAllReducer() if dist is not None else None
            if self.reducer is not None:
                self.reducer.always = force_dist  # one-rank functional check: still issue the RCCL all-reduces
                self.reducer.attach(self.model.model)  # all-reduces start from inside the backward, as gradient ranges become final
            # "16-mixed" = the reference's production precision: fp16 storage / operands under torch's GradScaler (FlatAdam
            # unscales, checks for inf / nan and skips on the device)
            self.scaler = torch.amp.GradScaler("cuda") if precision == "16-mixed" else None

    run = Run()
    model, opt = run.model, run.opt
    def make_label(image, g):
        if args.labels == "uniform":
            return torch.rand(image.shape[0], 1, c["H"], c["W"], device="cuda", generator=g)
        return learnable_label(image, generator=g)

    def make_batch(scaling):
        if scaling == "strong":
            # the global batch, identical on every rank, sharded by rank (ddp.shard_batch: rows [r*B, (r+1)*B))
            from mimo_unet_amd.ddp import shard_batch
            g = torch.Generator(device="cuda").manual_seed(100)
            image = torch.rand(c["batch"], c["Ci"], c["H"], c["W"], device="cuda", generator=g)
            b = shard_batch({"image": image, "label": make_label(image, g)}, rank, world)
            return {k: v.contiguous() for k, v in b.items()}
        g = torch.Generator(device="cuda").manual_seed(100 + rank)
        image = torch.rand(c["batch"], c["Ci"], c["H"], c["W"], device="cuda", generator=g)
        return {"image": image, "label": make_label(image, g)}

    batch = make_batch(args.scaling)
    feed = prefetcher = None

    def host_feed(resident):
        """(prefetcher, iterator): the same batch, but handed over from host memory every step"""
        import itertools
        from mimo_unet_amd.data import DevicePrefetcher
        host = {k: (v.cpu().pin_memory() if args.host_batches == "pinned" else v.cpu()) for k, v in resident.items()}
        pf = DevicePrefetcher(itertools.repeat(host), device=torch.device("cuda", dev), depth=int(os.environ.get("MIMO_PREFETCH_DEPTH", "2")))
        return pf, iter(pf)

    if args.host_batches:
        prefetcher, feed = host_feed(batch)

    adam_events = []

    def step(i, time_adam=False, r=None):
        r = r or run
        r.opt.zero_grad()
        out = r.model.training_step(batch if feed is None else next(feed), i)
        (r.scaler.scale(out["loss"]) if r.scaler is not None else out["loss"]).backward()
        if r.reducer is not None:
            r.reducer.finish()  # bucketed sums over RCCL were started from inside backward; FlatAdam scales by 1/world
        if time_adam:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if r.scaler is not None:
            r.scaler.step(r.opt)
            r.scaler.update()
        else:
            r.opt.step()
        if time_adam:
            e1.record()
            adam_events.append((e0, e1))
        return out["loss"]

    host_loop_s = []

    def timed_pass(r=None):
        """--warmup untimed steps, then EXACTLY --steps steps between barrier + synchronize; max over ranks"""
        for i in range(args.warmup):
            step(i, r=r)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss = step(i, r=r)
        host_loop_s.append(time.perf_counter() - t0)  # the host has enqueued every step (it may run ahead of the GPU)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, float(loss.detach())

    # ---- timed region (the regime of --scaling): no instrumentation ----
    elapsed, final_loss = timed_pass()
    collectives = list(run.reducer.last_issued) if run.reducer is not None else []
    # host time to ENQUEUE one step on an idle GPU (outside the timed region): the launch path's share of a step
    # (a diagnostic like the second pass below: --profile-steps 0 skips both, so that traces hold exactly warmup + steps)
    host_ms = []
    for i in range(3 if args.profile_steps > 0 else 0):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(i)
        host_ms.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    host_enqueue_ms = min(host_ms) if host_ms else None
    plan = next(p for p in model.model._plans.values() if not p.inference_only)
    # ---- second pass: HIP events on the launch stream around every kernel class / resolution tier ----
    prof, tiers, kind_tiers, psteps = {}, [], {}, max(0, args.profile_steps)
    if psteps:
        plan.profile(True)
        for i in range(psteps):
            step(i, time_adam=True)
        torch.cuda.synchronize()
        prof, tiers, kind_tiers = plan.profile_read(), plan.profile_read_tiers(), plan.profile_read_kind_tiers()
        plan.profile(False)
    # ---- the same timed pass under the STRICT arithmetic: three bf16-pair MFMAs per product in the weight gradient
    # (MIMO_WGRAD_NP=3, rounds 1-4; ~1e-5 per product like the forward and the data gradient) instead of the default's two
    # fp16 MFMAs with the activation as one fp16 value — a second model and plan in this process, same protocol ----
    strict = None
    if precision == "split16" and os.environ.get("MIMO_WGRAD_NP") is None and not args.no_strict:
        os.environ["MIMO_WGRAD_NP"] = "3"  # read when the plan is created (the run's first step)
        try:
            r3 = Run()
            strict_elapsed, _ = timed_pass(r3)
        finally:
            del os.environ["MIMO_WGRAD_NP"]
        strict = (world * B * args.steps / strict_elapsed, strict_elapsed / args.steps * 1e3)
        del r3
        torch.cuda.empty_cache()
    # ---- N > 1: the OTHER regime too, from a second timed pass (SURVEY 8d/e contract cfg3 as 32 GLOBAL = strong;
    # Lightning DDP semantics = the batch per device = weak), so that one line carries both, labelled ----
    other = "strong" if args.scaling == "weak" else "weak"
    other_elapsed = None
    if world > 1 and not args.one_regime and not (other == "strong" and c["batch"] % world):
        batch = make_batch(other)
        if args.host_batches:
            prefetcher, feed = host_feed(batch)
        other_elapsed, _ = timed_pass()
    identical = None
    if dist is not None:
        # data-parallel invariant: the same initial parameters + the same summed gradients => bit-identical parameters
        flat = model.model.flat_parameters()
        ref = flat.clone()
        dist.broadcast(ref, 0)
        same = torch.tensor([1 if torch.equal(ref, flat) else 0], device="cuda", dtype=torch.int32)
        dist.all_reduce(same, op=dist.ReduceOp.MIN)
        identical = bool(same.item())
    if rank != 0:
        dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed
    per_gpu = {"weak": c["batch"], "strong": c["batch"] // world}
    regime = {args.scaling: (value, ms_per_step)}
    if other_elapsed is not None:
        regime[other] = (world * per_gpu[other] * args.steps / other_elapsed, other_elapsed / args.steps * 1e3)
    elif world == 1:
        regime[other] = regime[args.scaling]  # one GPU: both regimes are the same run
    mixed = precision in ("bf16-mixed", "16-mixed")
    tier_bytes, bytes_per_image = algorithmic_bytes_per_image(c, 2 if mixed else 4)
    weight_bytes = 3.0 * 4.0 * sum(cin * cout * k * k for _, cin, cout, _, _, k in conv_layers(c))
    step_bytes = bytes_per_image * B + weight_bytes
    line = {
        "metric": "train images/sec at 256x256, S=2, fbc=30" if args.config == "cfg3" else f"train images/sec ({args.config})",
        "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        # the same timed pass with the weight gradient on three bf16-pair MFMAs per product (fp32-class like the rest of the
        # step; MIMO_WGRAD_NP=3) — the fence around the default's two-MFMA weight gradient
        "value_strict": None if strict is None else round(strict[0], 2),
        "ms_per_step_strict": None if strict is None else round(strict[1], 3),
        "dtype": {"fp32": "f32", "split16": "f32 (split into 16-bit hi/lo pairs on the MFMA; weight gradient: activation as one fp16 value)",
                  "bf16": "bf16 MFMA operands, f32 accumulate/storage (reduced precision)",
                  "bf16-mixed": "bf16 storage + MFMA operands, f32 accumulate / master weights / statistics (reduced precision)",
                  "16-mixed": "fp16 storage + MFMA operands under a loss scaler, f32 accumulate / master weights / statistics "
                              "(the reference's precision=16-mixed; reduced precision)"}.get(precision, precision),
        "data": "synthetic",
        "config": {"workload": f"{c['name']}, batch {c['batch']} " + ("global" if args.scaling == "strong" else "per GPU"),
                   "global_batch": world * B, "per_gpu_batch": B, "image": [c["H"], c["W"]], "scaling": args.scaling,
                   "labels": ("learnable_label(image): 5x5-smoothed channel mix + U(-0.025, 0.025) noise" if args.labels == "learnable"
                              else "U[0,1) (SURVEY 8d)") + "; same batch every step",
                   "parallelism": f"dp{world}" + ("" if dist is None else f" ({backend}: bucketed all-reduce started inside the backward)"),
                   "world_size": world, "backend": None if dist is None else backend,
                   # ranks that really ran RCCL (None under the gloo functional check and at one rank)
                   "rccl_ranks": dist.get_world_size() if dist is not None and backend == "nccl" else None,
                   # the gradient exchange of one step of the timed regime: (begin, end) float ranges of the flat buffer,
                   # in the order the backward issued them
                   "ddp_algorithm": None if run.reducer is None else run.reducer.algorithm,
                   "collectives_per_step": None if run.reducer is None else len(collectives),
                   "collective_mbytes": None if run.reducer is None else [round(4e-6 * (e - b), 2) for b, e in collectives],
                   "rank_devices": None if dist is None else devs,
                   # both data-parallel regimes, each from its own timed pass of --steps steps (value = the --scaling one)
                   "weak_images_per_s": round(regime["weak"][0], 2) if "weak" in regime else None,
                   "weak_ms_per_step": round(regime["weak"][1], 3) if "weak" in regime else None,
                   "weak_per_gpu_batch": per_gpu["weak"],
                   "strong_images_per_s": round(regime["strong"][0], 2) if "strong" in regime else None,
                   "strong_ms_per_step": round(regime["strong"][1], 3) if "strong" in regime else None,
                   "strong_global_batch": c["batch"],
                   "params_bit_identical_across_ranks": identical,
                   "optimizer": "adam(lr=1e-3) fused", "final_loss": round(final_loss, 5),
                   "inputs": ("resident in HBM" if prefetcher is None else
                              f"{args.host_batches} host tensors every step -> DevicePrefetcher(depth=2): "
                              f"the step waited for its worker {prefetcher.starved} times, {prefetcher.pageable_uploads} tensors from pageable memory"),
                   # host time to enqueue one step on an idle GPU (rank 0): well below ms_per_step = the GPU, not the launch path, bounds the step
                   "host_enqueue_ms_per_step": None if host_enqueue_ms is None else round(host_enqueue_ms, 3),
                   # host time of the TIMED loop per step, up to the point where every step was enqueued (below ms_per_step: the
                   # host ran ahead of the GPU; equal to it: the launch path — or a back-pressure wait — paces the step)
                   "host_loop_ms_per_step": round(host_loop_s[0] / args.steps * 1e3, 3),
                   "streams": ("weight gradients on the caller's stream" if os.environ.get("MIMO_WGRAD_STREAM") == "0" else
                               "weight gradients on a side stream beside the BatchNorm-backward kernels of the layer below "
                               "(MIMO_WGRAD_STREAM=1, default); the second pass that times the kernel classes serialises them")},
        # whole step against HBM: SURVEY 8(d) algorithmic bytes (activations + 3 x weights) / wall time / 8 TB/s
        "hbm_frac_step": round(step_bytes / (ms_per_step * 1e-3) / (HBM_PEAK_GBS * 1e9), 4),
    }
    if psteps:
        kernels = {}
        for name, r in prof.items():
            if r["launches"] == 0:
                continue
            sec = r["ms"] * 1e-3
            kernels[name] = {"avg_us": round(r["ms"] * 1e3 / r["launches"], 2), "launches_per_step": r["launches"] // psteps,
                             "ms_per_step": round(r["ms"] / psteps, 3), "tflops": round(r["flops"] / sec / 1e12, 2),
                             "algorithmic_gbs": round(r["bytes"] / sec / 1e9, 1),
                             "algorithmic_bytes_per_launch": int(r["bytes"] / r["launches"])}
        # bandwidth-class kernels (HBM roofline) are reported next to the convolution classes (MFMA roofline)
        bw_kernels = {k: kernels.pop(k) for k in list(kernels) if not k.startswith("conv3x3")}
        for v in bw_kernels.values():
            v["hbm_frac"] = round(v["algorithmic_gbs"] / HBM_PEAK_GBS, 4)
            del v["tflops"]
        if adam_events:
            ms = sum(a.elapsed_time(b) for a, b in adam_events) / len(adam_events)
            nparam = model.model.flat_parameters().numel()
            bw_kernels["adam"] = {"avg_us": round(ms * 1e3, 2), "launches_per_step": 1, "ms_per_step": round(ms, 3),
                                  "algorithmic_gbs": round(28.0 * nparam / (ms * 1e-3) / 1e9, 1),
                                  "algorithmic_bytes_per_launch": 28 * nparam,
                                  "hbm_frac": round(28.0 * nparam / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])

        def class_peak(kind):
            """dense 16-bit MFMA peak / MFMAs per algorithmic product of that class"""
            if precision == "fp32":
                return FP32_MFMA_PEAK_TFLOPS
            if precision == "split16":  # forward / data gradient: three MFMAs; weight gradient (round 5): two fp16 MFMAs
                return 2500.0 / 2.0 if kind == "conv3x3_wgrad" and os.environ.get("MIMO_WGRAD_NP") != "3" else SPLIT16_PEAK_TFLOPS
            return 2500.0

        for k, v in kernels.items():
            v["peak_tflops"] = round(class_peak(k), 1)
            v["frac"] = round(v["tflops"] / class_peak(k), 4)
        peak = class_peak(dom)
        traffic = step_traffic = None
        traffic_src = "counter summaries are collected for cfg3 / split16 / batch 32 on one GPU only"
        if args.config == "cfg3" and precision == "split16" and world == 1 and not args.batch:
            pt, traffic_src = pmc_traffic()
            if pt is not None:
                traffic = int(pt["classes"][dom]["bytes_per_launch"]) if dom in pt.get("classes", {}) else None
                step_traffic = pt.get("step_bytes")
        tier_ms = [(f + b) / psteps for f, b in tiers]
        line["roofline"] = {
            "kernel": dom, "bound": "mfma", "achieved": kernels[dom]["tflops"], "peak": round(peak, 1),
            "unit": "TFLOP/s", "frac": round(kernels[dom]["tflops"] / peak, 4),
            # the same flops against the raw dense 16-bit MFMA peak (no division by the 3 MFMAs split16 spends per product)
            "frac_raw_peak": round(kernels[dom]["tflops"] / (FP32_MFMA_PEAK_TFLOPS if precision == "fp32" else 2500.0), 4),
            "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_src,
            "step_traffic_bytes": step_traffic,
            "arithmetic": {"fp32": "f32-input MFMA",
                           "split16": "forward / data gradient: 3x 16-bit MFMA per product (fp16 hi/lo forward, bf16 hi/lo data "
                                      "gradient), peak = 2500 TFLOP/s dense 16-bit MFMA / 3; weight gradient: 2x fp16 MFMA per "
                                      "product (activation as one fp16 value x dz as an fp16 pair), peak = 2500 / 2; fp32 accumulate",
                           "bf16": "bf16 MFMA operands (one MFMA per product), fp32 accumulate and storage — reduced "
                                   "precision, NOT the fp32 metric",
                           "bf16-mixed": "bf16 storage and MFMA operands (one MFMA per product), fp32 accumulate — reduced "
                                         "precision, NOT the fp32 metric",
                           "16-mixed": "fp16 storage and MFMA operands (one MFMA per product), fp32 accumulate, loss "
                                       "scaling — reduced precision, NOT the fp32 metric"}.get(precision, precision),
            "source": f"second pass of {psteps} steps with HIP events on the launch stream (the timed region is not instrumented)",
            "hbm_frac_algorithmic": round(kernels[dom]["algorithmic_gbs"] / HBM_PEAK_GBS, 4),
            "conv_ms_per_step": round(sum(k["ms_per_step"] for k in kernels.values()), 2), "kernels": kernels,
            # per resolution tier: algorithmic activation bytes of the tier's conv layers (train = 3 x
            # forward) / summed device time of EVERY kernel launched for that tier's blocks / 8 TB/s
            "tiers": {f"{c['H'] >> t}x{c['W'] >> t}": {
                "ms_per_step": round(tier_ms[t], 3), "algorithmic_mb_per_image": round(tier_bytes[t] / 1e6, 1),
                "hbm_frac": round(tier_bytes[t] * B / (tier_ms[t] * 1e-3) / (HBM_PEAK_GBS * 1e9), 4) if tier_ms[t] > 0 else None,
                # which kernel classes the tier's time is made of (ms per step); "other" = statistics / reduction /
                # packing launches that carry no class of their own
                "kernels_ms": dict({k: round(v[t] / psteps, 3) for k, v in kind_tiers.items() if v[t] > 0},
                                   other=round(tier_ms[t] - sum(v[t] for v in kind_tiers.values()) / psteps, 3))}
                for t in range(5)},
            # the bandwidth class, priced against HBM (8000 GB/s): algorithmic bytes / HIP-event time
            "bandwidth_kernels": {"peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "ms_per_step": round(sum(k["ms_per_step"] for k in bw_kernels.values()), 2),
                                  "kernels": bw_kernels}}
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(c, labels=args.labels)
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
