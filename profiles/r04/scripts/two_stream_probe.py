#!/usr/bin/env python3
"""Probe: how much would kernel-level concurrency between INDEPENDENT chains buy?  Two independent cfg3 models (each at half
the batch: 16 images, together the headline batch 32) train on two HIP streams at once; compared with the same two models
stepping one after the other on one stream, and with one model at batch 32.  The per-subnetwork chains of the encoder and
of the decoder heads are independent in exactly this way (model.py:150-175, 232-243), so this bounds what running them on
two streams could gain: an MFMA-bound kernel of one chain next to a bandwidth-bound kernel of the other."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import CONFIGS, learnable_label, make_model  # noqa: E402


def make(batch, seed):
    m = make_model(CONFIGS["cfg3"]).cuda()
    m.train()
    g = torch.Generator(device="cuda").manual_seed(100 + seed)
    image = torch.rand(batch, 2, 256, 256, device="cuda", generator=g)
    return m, m.configure_optimizers()["optimizer"], {"image": image, "label": learnable_label(image, generator=g)}


def step(m, opt, b, i):
    opt.zero_grad()
    m.training_step(b, i)["loss"].backward()
    opt.step()


def timed(fn, steps=20, warm=6):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


A, B, C = make(16, 0), make(16, 1), make(32, 2)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def serial(i):
    step(*A, i)
    step(*B, i)


def concurrent(i):
    with torch.cuda.stream(sa):
        step(*A, i)
    with torch.cuda.stream(sb):
        step(*B, i)


for rep in range(3):
    t1 = timed(lambda i: step(*C, i))
    t2 = timed(serial)
    t3 = timed(concurrent)
    print(f"one model, batch 32: {t1:7.3f} ms   two models x batch 16, one stream: {t2:7.3f} ms   two streams: {t3:7.3f} ms "
          f"({(t2 / t3 - 1) * 100:+.1f} % vs one stream, {(t1 / t3 - 1) * 100:+.1f} % vs batch 32)")
