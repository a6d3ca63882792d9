"""Where the HOST time of a steady-state training step goes (cProfile over the bench loop, GPU busy):
    python scripts/host_profile.py [batch=4] [steps=200] [ddp=0|1]
Prints the host loop time per step next to the GPU time per step, then the functions by cumulative and by own time."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench as B

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ddp = len(sys.argv) > 3 and sys.argv[3] == "1"
c = dict(B.CONFIGS["cfg3"], batch=batch)
torch.cuda.set_device(0)
red = None
if ddp:
    import torch.distributed as dist
    from mimo_unet_amd.ddp import FlatGradientAllReducer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.manual_seed(1)
model = B.make_model(c).cuda()
model.train()
opt = model.configure_optimizers()["optimizer"]
if ddp:
    red = FlatGradientAllReducer()
    red.always = True
    red.attach(model.model)
g = torch.Generator(device="cuda").manual_seed(100)
image = torch.rand(batch, c["Ci"], c["H"], c["W"], device="cuda", generator=g)
b = {"image": image, "label": torch.rand(batch, 1, c["H"], c["W"], device="cuda", generator=g)}


def step(i):
    opt.zero_grad()
    out = model.training_step(b, i)
    out["loss"].backward()
    if red is not None:
        red.finish()
    opt.step()


for i in range(20):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    step(i)
th = time.perf_counter() - t0
torch.cuda.synchronize()
tg = time.perf_counter() - t0
print(f"batch {batch} ddp {int(ddp)} graph {os.environ.get('MIMO_TRAIN_GRAPH', '0')}: host loop {th / steps * 1e3:.3f} ms/step, "
      f"with the GPU drained {tg / steps * 1e3:.3f} ms/step")
# the backward runs on autograd's own thread (cProfile does not see it): time its layers by hand
from mimo_unet_amd.engine import Plan
from mimo_unet_amd.models.mimo_components import model as M
acc = {"_NetFunction.backward": 0.0, "MimoUNet._run_backward": 0.0, "Plan.backward (C calls)": 0.0, "grad_ready_hook": 0.0}


def timed(fn, key):
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[key] += time.perf_counter() - t
    return w


M._NetFunction.backward = staticmethod(timed(M._NetFunction.backward, "_NetFunction.backward"))
M.MimoUNet._run_backward = timed(M.MimoUNet._run_backward, "MimoUNet._run_backward")
Plan.backward = timed(Plan.backward, "Plan.backward (C calls)")
if model.model.grad_ready_hook is not None:
    model.model.grad_ready_hook = timed(model.model.grad_ready_hook, "grad_ready_hook")
t0 = time.perf_counter()
tb = 0.0
for i in range(steps):
    opt.zero_grad()
    out = model.training_step(b, i)
    t = time.perf_counter()
    out["loss"].backward()
    tb += time.perf_counter() - t
    if red is not None:
        red.finish()
    opt.step()
torch.cuda.synchronize()
print(f"loss.backward() {tb / steps * 1e3:.3f} ms per step on the host, of which: " +
      ", ".join(f"{k} {v / steps * 1e3:.3f}" for k, v in acc.items()))
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    step(i)
pr.disable()
torch.cuda.synchronize()
for key in ("cumulative", "tottime"):
    print(f"---- by {key} (ms per step = total / {steps}) ----")
    st = pstats.Stats(pr)
    st.sort_stats(key)
    rows = []
    for func, (cc, nc, tt, ct, callers) in st.stats.items():
        rows.append((ct if key == "cumulative" else tt, nc, tt, ct, func))
    rows.sort(reverse=True)
    for v, nc, tt, ct, func in rows[:45]:
        print(f"{nc / steps:7.1f} calls  own {tt / steps * 1e3:7.3f} ms  cum {ct / steps * 1e3:7.3f} ms  {os.path.basename(func[0])}:{func[1]} {func[2]}")
if ddp:
    dist.destroy_process_group()
