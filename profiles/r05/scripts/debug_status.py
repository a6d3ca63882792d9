import sys
sys.path.insert(0, ".")
import torch
from tests.helpers import cfg_from_meta, load_npz, state_from
from tests.test_network_gpu import build_model
fx = load_npz("mini_s2_step.npz")
cfg = cfg_from_meta(fx["meta"])
model = build_model(cfg, state_from(fx, "init/"))
x = torch.rand(3, 2, 32, 32, generator=torch.Generator().manual_seed(3)).cuda()
lab = torch.rand(3, 1, 32, 32, generator=torch.Generator().manual_seed(4)).cuda()
model.train()
model.training_step({"image": x, "label": lab}, 0)
w = dict(model.model.named_parameters())["core.down2.conv.double_conv.1.weight"]
with torch.no_grad():
    w[0] = 3.0e5
out = model.training_step({"image": x, "label": lab}, 0)
print("train status", model.model.numerics_status(), float(out["loss"]))
sd = model.state_dict()
for k in ("model.core.down2.conv.double_conv.1.running_var", "model.core.down2.conv.double_conv.4.running_var", "model.core.down2.conv.double_conv.4.running_mean"):
    print(k, sd[k][:4].tolist())
model.eval()
x5 = x[:, None].repeat(1, 2, 1, 1, 1)
with torch.no_grad():
    p1, p2 = model(x5)
print("eval status", model.model.numerics_status(), bool(torch.isfinite(p1).all()), float(p1.abs().max()))
