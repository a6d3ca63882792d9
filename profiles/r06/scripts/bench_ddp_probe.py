"""Where does the host go in bench.py's one-rank RCCL route?  Timers around the C calls and the reducer (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mimo_unet_amd.engine import Plan
from mimo_unet_amd.ddp import FlatGradientAllReducer
acc, cnt = {}, {}
def timed(cls, name):
    fn = getattr(cls, name)
    key = f"{cls.__name__}.{name}"
    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[key] = acc.get(key, 0.0) + time.perf_counter() - t
            cnt[key] = cnt.get(key, 0) + 1
    setattr(cls, name, w)
for n in ("forward", "backward", "loss_forward", "bind"):
    timed(Plan, n)
for n in ("start", "finish", "_issue"):
    timed(FlatGradientAllReducer, n)
import torch.distributed as dist
_ar = dist.all_reduce
def ar(*a, **k):
    t = time.perf_counter()
    try:
        return _ar(*a, **k)
    finally:
        acc["dist.all_reduce"] = acc.get("dist.all_reduce", 0.0) + time.perf_counter() - t
        cnt["dist.all_reduce"] = cnt.get("dist.all_reduce", 0) + 1
dist.all_reduce = ar
sys.argv = ["bench.py", "--batch", "4", "--steps", "200", "--warmup", "15", "--no-cpu-baseline", "--no-strict", "--profile-steps", "0"]
import bench
_t0 = time.perf_counter()
try:
    bench.main()
except SystemExit:
    pass
print("wall", time.perf_counter() - _t0)
for k in sorted(acc):
    print(f"{k:36s} {cnt[k]:6d} calls  {acc[k] / cnt[k] * 1e3:8.3f} ms per call  {acc[k] * 1e3 / 215:8.3f} ms per step")
# which threads of this process burned CPU time?
import glob
rows = []
for d in glob.glob("/proc/self/task/*"):
    try:
        comm = open(d + "/comm").read().strip()
        f = open(d + "/stat").read().rsplit(")", 1)[1].split()
        rows.append(((int(f[11]) + int(f[12])) / os.sysconf("SC_CLK_TCK"), comm, os.path.basename(d)))
    except OSError:
        pass
for cpu, comm, tid in sorted(rows, reverse=True)[:12]:
    print(f"thread {tid:>8s} {comm:20s} cpu {cpu:7.2f} s")
print("threads:", len(rows))
