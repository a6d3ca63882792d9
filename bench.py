#!/usr/bin/env python3
"""Headline benchmark: MIMO U-Net training throughput (images/s) at 256x256, S=2, fbc=30.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One "step" = what Lightning runs per batch for the reference's `MimoUnetModel`
(mimo/models/mimo_unet.py:115-144 + backward + Adam): draw the S batch permutations,
forward (gather fused into the first kernel), Laplace NLL, loss-buffer weighting, backward,
[gradient all-reduce over RCCL when N>1], fused Adam.  Synthetic U[0,1) inputs, PyTorch
default random init, fp32 end to end.  Weak scaling: the per-GPU batch is fixed (32, the
reference README's SEN12TP batch size; Lightning DDP semantics = per-device batch).

Prints ONE JSON line on rank 0 with the throughput, the roofline of the dominant kernel class
(device time from HIP events recorded around every launch inside the timed region) and — at
N=1 — the CPU oracle timed on this box's host cores for the same workload shape.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE.json configs[2] geometry; per-GPU batch 32 (README: --batch_size 32)
    "cfg3": dict(name="cfg3: SEN12TP-shape synthetic 2->1 ch, 256x256, S=2, fbc=30, laplace_nll, batch 32 per GPU",
                 Ci=2, Co=2, S=2, f=30, H=256, W=256, batch=32),
    # BASELINE.json configs[1]
    "cfg2": dict(name="cfg2: NYUv2-shape synthetic 3->1 ch, 256x256, S=2, fbc=21, laplace_nll, batch 64 per GPU",
                 Ci=3, Co=2, S=2, f=21, H=256, W=256, batch=64),
    # BASELINE.json configs[3] geometry (S=4 head-width stress; 128 global / 8 GPUs); run with MIMO_PRECISION=bf16
    # for its arithmetic
    "cfg4": dict(name="cfg4: synthetic 2->1 ch, 256x256, S=4, fbc=30, laplace_nll, batch 16 per GPU",
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


def make_model(c):
    from mimo.models.mimo_unet import MimoUnetModel
    return MimoUnetModel(in_channels=c["Ci"], out_channels=c["Co"], num_subnetworks=c["S"], filter_base_count=c["f"],
                         center_dropout_rate=0.0, final_dropout_rate=0.0, encoder_dropout_rate=0.0, core_dropout_rate=0.0,
                         decoder_dropout_rate=0.0, loss="laplace_nll", weight_decay=0.0, learning_rate=1e-3, seed=1,
                         loss_buffer_size=10, loss_buffer_temperature=0.3)


def pmc_traffic(kernel_class):
    """HBM bytes per launch of a kernel class from the committed counter passes of this same command
    (profiles/<round>/final/pmc_traffic.json, written from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs —
    counters cannot be collected from inside the timed run).  None when no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "final", "pmc_traffic.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as fh:
            t = json.load(fh)
        return int(t["classes"][kernel_class]["bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except (KeyError, ValueError, OSError):
        return None, None


def cpu_baseline(c, batch=4, steps=2, threads=None):
    """The CPU oracle (restatement of the reference's step, pinned to reference goldens) on the
    host cores of this box: same network shape, bounded sample.  Thread count: torch's CPU
    convolutions stop scaling (and collapse when oversubscribed) well below this box's core
    count (measured on the 256-core GPU box: 8 thr 2.6, 16 thr 3.4, 32 thr 3.0, 64 thr 1.7, 128 thr 0.8
    images/s), so the run uses min(cores, MIMO_BENCH_CPU_THREADS or 16) threads and reports that."""
    from oracle import mimo_oracle as O
    threads = threads or min(os.cpu_count() or 1, int(os.environ.get("MIMO_BENCH_CPU_THREADS", "16")))
    torch.set_num_threads(threads)
    cfg = O.NetConfig(c["Ci"], c["Co"], c["S"], c["f"])
    ts = O.TrainState(cfg=cfg, st=O.init_state(cfg, 1), loss_buffer=O.LossBuffer(c["S"], 0.3, 10))
    g = torch.Generator().manual_seed(1)
    image = torch.rand(batch, c["Ci"], c["H"], c["W"], generator=g)
    label = learnable_label(image, generator=g)
    times = []
    for i in range(steps + 1):
        perms = O.draw_perms(batch, c["S"], generator=g)
        t0 = time.perf_counter()
        O.train_step(ts, image, label, None, perms)
        times.append(time.perf_counter() - t0)
    dt = sum(times[1:]) / steps
    return {"value": round(batch / dt, 3), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} timed steps (1 warm-up) of the same network shape at batch {batch}, fp32, "
                      f"torch {torch.__version__} CPU, {dt * 1e3:.0f} ms/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg3", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    c = dict(CONFIGS[args.config])
    if args.batch:
        c["name"] = c["name"].replace(f"batch {c['batch']} per GPU", f"batch {args.batch} per GPU (non-default)")
        c["batch"] = args.batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # one process per GPU over RCCL ("nccl").  MIMO_BENCH_BACKEND=gloo lets several ranks share one GPU
    # (functional check of the data-parallel path on a single-GPU box; not a performance configuration)
    backend = os.environ.get("MIMO_BENCH_BACKEND", "nccl")
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

    torch.manual_seed(1)
    model = make_model(c).cuda()
    model.train()
    opt = model.configure_optimizers()["optimizer"]
    opt.grad_scale = 1.0 / world
    g = torch.Generator(device="cuda").manual_seed(100 + rank)
    B = c["batch"]
    image = torch.rand(B, c["Ci"], c["H"], c["W"], device="cuda", generator=g)
    # targets the image predicts (see learnable_label): with all-noise labels the NLL's scale head collapses on
    # single pixels and the loss spikes (reproduced by the fp64 CPU oracle, tests/tools/diag_trained_state.py)
    label = learnable_label(image, generator=g)
    batch = {"image": image, "label": label}

    from mimo_unet_amd.ddp import FlatGradientAllReducer
    reducer = FlatGradientAllReducer() if dist is not None else None
    if reducer is not None:
        reducer.always = force_dist  # one-rank functional check: still issue the RCCL all-reduces
        reducer.attach(model.model)  # all-reduce of the core/decoder gradients overlaps the encoder backward

    def step(i):
        opt.zero_grad()
        out = model.training_step(batch, i)
        out["loss"].backward()
        if reducer is not None:
            reducer.finish()  # bucketed sums over RCCL were started from inside backward; FlatAdam scales by 1/world
        opt.step()
        return out["loss"]

    for i in range(args.warmup):
        step(i)
    plan = next(iter(model.model._plans.values()))
    plan.profile(True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof = plan.profile_read()
    plan.profile(False)
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed
    kernels = {}
    for name, r in prof.items():
        if r["launches"] == 0:
            continue
        sec = r["ms"] * 1e-3
        kernels[name] = {"avg_us": round(r["ms"] * 1e3 / r["launches"], 2), "launches_per_step": r["launches"] // args.steps,
                         "ms_per_step": round(r["ms"] / args.steps, 3), "tflops": round(r["flops"] / sec / 1e12, 2),
                         "algorithmic_gbs": round(r["bytes"] / sec / 1e9, 1),
                         "algorithmic_bytes_per_launch": int(r["bytes"] / r["launches"])}
    # bandwidth-class kernels (HBM roofline) are reported next to the convolution classes (MFMA roofline)
    bw_kernels = {k: kernels.pop(k) for k in list(kernels) if not k.startswith("conv3x3")}
    for v in bw_kernels.values():
        v["hbm_frac"] = round(v["algorithmic_gbs"] / HBM_PEAK_GBS, 4)
        del v["tflops"]
    dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
    precision = os.environ.get("MIMO_PRECISION", "split16")
    peak = {"fp32": FP32_MFMA_PEAK_TFLOPS, "split16": SPLIT16_PEAK_TFLOPS, "bf16": 2500.0}[precision]
    traffic, traffic_src = pmc_traffic(dom) if (args.config == "cfg3" and precision == "split16") else (None, None)
    roofline = {"kernel": dom, "bound": "mfma", "achieved": kernels[dom]["tflops"], "peak": round(peak, 1),
                "unit": "TFLOP/s", "frac": round(kernels[dom]["tflops"] / peak, 4), "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_src,
                "arithmetic": {"fp32": "f32-input MFMA",
                               "split16": "3x 16-bit MFMA per product (fp16 hi/lo forward, bf16 hi/lo gradients), fp32 "
                                          "accumulate; peak = 2500 TFLOP/s dense 16-bit MFMA / 3",
                               "bf16": "bf16 MFMA operands (one MFMA per product), fp32 accumulate and storage — reduced "
                                       "precision, NOT the fp32 metric"}[precision],
                "hbm_frac_algorithmic": round(kernels[dom]["algorithmic_gbs"] / HBM_PEAK_GBS, 4),
                "conv_ms_per_step": round(sum(k["ms_per_step"] for k in kernels.values()), 2), "kernels": kernels,
                # the bandwidth class, priced against HBM (8000 GB/s): algorithmic bytes (8 or 12 B per element of the
                # BatchNorm passes) / HIP-event time of the same timed region
                "bandwidth_kernels": {"peak": HBM_PEAK_GBS, "unit": "GB/s", "kernels": bw_kernels}}
    line = {
        "metric": "train images/sec at 256x256, S=2, fbc=30" if args.config == "cfg3" else f"train images/sec ({args.config})",
        "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp32": "f32", "split16": "f32 (split into 16-bit hi/lo pairs on the MFMA)",
                  "bf16": "bf16 MFMA operands, f32 accumulate/storage (reduced precision)"}[precision],
        "data": "synthetic",
        "config": {"workload": c["name"], "global_batch": world * B, "per_gpu_batch": B, "image": [c["H"], c["W"]],
                   "parallelism": f"dp{world}" + ("" if world == 1 else f" ({backend}, all-reduce overlapped with the encoder backward)"),
                   "optimizer": "adam(lr=1e-3) fused", "final_loss": round(float(loss.detach()), 5)},
        "roofline": roofline,
    }
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(c)
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
