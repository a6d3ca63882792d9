"""How does the CPU oracle's training step scale with threads on this box? (diagnostic)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import mimo_oracle as O
cfg = O.NetConfig(2, 2, 2, 30)
print("cpu_count", os.cpu_count())
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    ts = O.TrainState(cfg=cfg, st=O.init_state(cfg, 1), loss_buffer=O.LossBuffer(2, 0.3, 10))
    g = torch.Generator().manual_seed(1)
    image, label = torch.rand(4, 2, 256, 256, generator=g), torch.rand(4, 1, 256, 256, generator=g)
    perms = O.draw_perms(4, 2, generator=g)
    t0 = time.perf_counter(); O.train_step(ts, image, label, None, perms); t1 = time.perf_counter()
    O.train_step(ts, image, label, None, perms); t2 = time.perf_counter()
    print(th, "threads: first %.2fs second %.2fs -> %.3f img/s" % (t1 - t0, t2 - t1, 4 / (t2 - t1)), flush=True)
