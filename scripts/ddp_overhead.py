"""What the data-parallel step costs beside the plain step, on one GPU (1-rank RCCL group):
    plain | staged backward with a no-op hook | + bucketed all-reduces (reducer.always) | + finish() before the optimiser
python scripts/ddp_overhead.py [batch=32]      (MIMO_TRAIN_GRAPH=0|1 in the environment selects eager launches / graph replay)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench as B
from mimo_unet_amd.ddp import FlatGradientAllReducer

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
c = dict(B.CONFIGS["cfg3"])
if len(sys.argv) > 1:
    c["batch"] = int(sys.argv[1])
g = torch.Generator(device="cuda").manual_seed(100)
image = torch.rand(c["batch"], c["Ci"], c["H"], c["W"], device="cuda", generator=g)
batch = {"image": image, "label": B.learnable_label(image, generator=g)}

def run(name, setup):
    torch.manual_seed(1)
    model = B.make_model(c).cuda(); model.train()
    opt = model.configure_optimizers()["optimizer"]
    red = setup(model)
    def step(i):
        opt.zero_grad()
        model.training_step(batch, i)["loss"].backward()
        if red is not None: red.finish()
        opt.step()
    n = 30 if c["batch"] >= 16 else 100
    for i in range(8): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): step(i)
    th = time.perf_counter() - t0  # the host has enqueued everything
    torch.cuda.synchronize()
    print(f"{name:48s} {(time.perf_counter() - t0) / n * 1e3:7.3f} ms/step   (host loop {th / n * 1e3:6.3f} ms/step)", flush=True)
    del model, opt

def noop_hook(m):
    m.model.grad_ready_hook = lambda flat, b, e: None
    return None
def reducer(always):
    def f(m):
        r = FlatGradientAllReducer(); r.always = always; r.attach(m.model); return r
    return f
run("plain, before the process group exists", lambda m: None)
run("plain, before the process group exists", lambda m: None)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
def with_async(flag, setup):
    def f(m):
        m.model.async_stages = flag
        return setup(m)
    return f
for rep in range(2):
    run("plain", lambda m: None)
    for flag in (False, True):
        tag = "async stages" if flag else "joined stages"
        run(f"staged backward ({tag}), no-op hook", with_async(flag, noop_hook))
        run(f"staged ({tag}) + 1-rank RCCL all-reduces", with_async(flag, reducer(True)))
dist.destroy_process_group()
