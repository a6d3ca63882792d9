"""Where does the host block inside a steady-state training step?  python scripts/host_phases.py [batch] — host time of
zero_grad / training_step / backward / optimizer.step per step with the GPU busy, against the GPU time per step."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench as B
c = dict(B.CONFIGS["cfg3"]); c["batch"] = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(1)
model = B.make_model(c).cuda(); model.train()
opt = model.configure_optimizers()["optimizer"]
g = torch.Generator(device="cuda").manual_seed(100)
image = torch.rand(c["batch"], c["Ci"], c["H"], c["W"], device="cuda", generator=g)
batch = {"image": image, "label": B.learnable_label(image, generator=g)}
def step(i, T):
    t0 = time.perf_counter(); opt.zero_grad()
    t1 = time.perf_counter(); out = model.training_step(batch, i)
    t2 = time.perf_counter(); out["loss"].backward()
    t3 = time.perf_counter(); opt.step()
    t4 = time.perf_counter()
    T.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
T = []
for i in range(10): step(i, T)
torch.cuda.synchronize(); T = []
t0 = time.perf_counter()
for i in range(30): step(i, T)
th = time.perf_counter() - t0
torch.cuda.synchronize(); tg = time.perf_counter() - t0
import numpy as np
A = np.array(T) * 1e3
print("batch", c["batch"], "host loop ms/step %.2f  gpu ms/step %.2f" % (th / 30 * 1e3, tg / 30 * 1e3))
print("median ms  zero_grad %.3f  training_step %.3f  backward %.3f  opt.step %.3f" % tuple(np.median(A, axis=0)))
print("first 6 steps:", np.round(A[:6], 2).tolist())
