import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import torch.distributed as dist
import bench as B
from mimo_unet_amd.ddp import FlatGradientAllReducer
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
torch.cuda.set_device(0)
c = dict(B.CONFIGS["cfg3"], batch=4)
g = torch.Generator(device="cuda").manual_seed(100)
image = torch.rand(4, 2, 256, 256, device="cuda", generator=g)
batch = {"image": image, "label": torch.rand(4, 1, 256, 256, device="cuda", generator=g)}
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
def run(name, bcast=False, barrier=False, gather=False, contig=False, keep=False, scale=False):
    if gather:
        devs = [None]; dist.all_gather_object(devs, "x")
    torch.manual_seed(1)
    model = B.make_model(c).cuda(); model.train()
    if bcast:
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, 0)
    opt = model.configure_optimizers()["optimizer"]
    red = FlatGradientAllReducer(); red.always = True; red.attach(model.model)
    b = {k: v.contiguous() for k, v in batch.items()} if contig else batch
    if scale:
        opt.reduce_scale = 1.0 / 1
    held = {}
    def step(i):
        opt.zero_grad()
        out = model.training_step(b, i)
        out["loss"].backward()
        red.finish()
        opt.step()
        if keep:
            held["out"] = out
        return out["loss"]
    for i in range(15): step(i)
    if barrier: dist.barrier()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(100):
        loss = step(i)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"{name:40s} {(time.perf_counter() - t0) * 10:7.3f} ms/step (host loop {th * 10:6.3f})", flush=True)
run("baseline")
if os.environ.get("BISECT_ONLY") == "1":
    dist.destroy_process_group()
    sys.exit(0)
if os.environ.get("BISECT_PROPS", "1") == "1":
    n = torch.cuda.device_count()
    p = torch.cuda.get_device_properties(0)
    print("device_count", n, "properties:", getattr(p, "pci_bus_id", None), getattr(p, "uuid", None), flush=True)
    run("after torch.cuda.get_device_properties")
run("+ broadcast of every tensor at start", bcast=True)
run("+ barrier in front of the loop", barrier=True)
run("+ all_gather_object", gather=True)
run("baseline again")
run("step outputs kept alive until the next step (bench.py)", keep=True)
run("+ reduce_scale set", keep=True, scale=True)
run("all three", bcast=True, barrier=True, gather=True)
dist.destroy_process_group()
