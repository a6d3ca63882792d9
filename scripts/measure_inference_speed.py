#!/usr/bin/env python3
"""Inference latency / throughput of the (MC-dropout) ensemble — same harness shape as the
reference's scripts/test/measure_inference_speed.py:22-47 (10 warm-up + N timed forward passes
of `EnsembleModule` bracketed by device events, mean / std in ms), runnable without a checkpoint:
by default it builds a randomly initialised model of BASELINE config 5 (3->1 ch, S=1, fbc=30,
encoder/core/decoder dropout 0.1, 16 Monte-Carlo passes, 256x256).

    python scripts/measure_inference_speed.py [--model_checkpoint_paths a.ckpt b.ckpt] \
        [--monte_carlo_steps 16] [--batch 1] [--height 256 --width 256] [--repetitions 200]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mimo.models.ensemble import EnsembleModule  # noqa: E402
from mimo.models.mimo_unet import MimoUnetModel  # noqa: E402


def algorithmic_work(net, H, W):
    """(conv flops, HBM bytes) of ONE forward pass of one image on the inference fast path: 2*9*Cin*Cout per output pixel of
    every 3x3 convolution; bytes = every convolution reads its input and writes its (BatchNorm + ReLU fused) output once in
    fp32, plus the MaxPool2d / up-sample + concat passes and the 1x1 head (materialised-tensor accounting, as SURVEY 8d)."""
    div = {"in_convs": 1, "down1s": 2, "down2": 4, "down3": 8, "down4": 16, "up1": 8, "up2": 4, "up3": 2, "up4s": 1}
    flops = byts = 0.0
    for name, w in net.state_dict().items():
        if w.dim() != 4:
            continue
        co, ci, kh, kw = w.shape
        d = next((v for k, v in div.items() if f".{k}." in f".{name}"), 1)
        px = (H // d) * (W // d)
        if kh == 3:
            flops += 2.0 * 9 * ci * co * px
        byts += 4.0 * (ci + co) * px
        if kh == 3 and ".double_conv.0." in name and d > 1 and "up" not in name:
            byts += 4.0 * ci * px * 5  # its input came out of a 2x2 max-pool: read 4, write 1 per pooled element
        if kh == 3 and ".double_conv.0." in name and "up" in name:
            byts += 4.0 * ci * px * 1.25  # ... or out of up-sample + concat: read skip + quarter-size tensor, write concat
    return flops, byts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model_checkpoint_paths", nargs="*", default=[])
    ap.add_argument("--monte_carlo_steps", type=int, default=16)
    ap.add_argument("--in_channels", type=int, default=3)
    ap.add_argument("--num_subnetworks", type=int, default=1)
    ap.add_argument("--filter_base_count", type=int, default=30)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--repetitions", type=int, default=200)
    args = ap.parse_args()

    if args.model_checkpoint_paths:
        model = EnsembleModule(args.model_checkpoint_paths, monte_carlo_steps=args.monte_carlo_steps, keep_on_device=True)
    else:
        torch.manual_seed(0)
        m = MimoUnetModel(in_channels=args.in_channels, out_channels=2, num_subnetworks=args.num_subnetworks,
                          filter_base_count=args.filter_base_count, center_dropout_rate=0.0, final_dropout_rate=0.0,
                          encoder_dropout_rate=args.dropout, core_dropout_rate=args.dropout,
                          decoder_dropout_rate=args.dropout, loss="laplace_nll", weight_decay=0.0, learning_rate=1e-3,
                          seed=0, loss_buffer_size=10, loss_buffer_temperature=0.3)
        model = EnsembleModule([], monte_carlo_steps=args.monte_carlo_steps, models=[m], keep_on_device=True)
    model.cuda()
    x = torch.randn(args.batch, args.in_channels, args.height, args.width, device="cuda")
    for _ in range(10):
        model(x)
    starter, ender = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = np.zeros(args.repetitions)
    with torch.no_grad():
        for i in range(args.repetitions):
            starter.record()
            model(x)
            ender.record()
            torch.cuda.synchronize()
            t[i] = starter.elapsed_time(ender)
    passes = max(1, args.monte_carlo_steps)
    fl, by = algorithmic_work(model.models[0].model, args.height, args.width)
    n_pass = args.batch * passes
    sec = float(t.mean()) * 1e-3
    # peaks: 2500 TFLOP/s dense 16-bit MFMA / 3 MFMAs per product (split16 default), HBM 8 TB/s (MI355X_MICROARCH.md)
    roof = {"conv_gflop_per_pass": round(fl / 1e9, 3), "achieved_tflops": round(fl * n_pass / sec / 1e12, 1),
            "mfma_frac": round(fl * n_pass / sec / 833.3e12, 4), "algorithmic_mb_per_pass": round(by / 1e6, 1),
            "achieved_gbs": round(by * n_pass / sec / 1e9, 1), "hbm_frac": round(by * n_pass / sec / 8e12, 4)}
    print(json.dumps({"roofline": roof, "what": "ensemble forward incl. uncertainty reduction, results left on device",
                      "batch": args.batch, "monte_carlo_steps": args.monte_carlo_steps, "image": [args.height, args.width],
                      "S": model.models[0].num_subnetworks, "fbc": args.filter_base_count,
                      "mean_ms": round(float(t.mean()), 3), "std_ms": round(float(t.std()), 3),
                      "images_per_s": round(args.batch / (t.mean() * 1e-3), 1),
                      "stochastic_forward_passes_per_s": round(args.batch * passes / (t.mean() * 1e-3), 1)}))


if __name__ == "__main__":
    main()
