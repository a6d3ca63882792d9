"""Diagnostic: HIP (fp32) and oracle (fp32) gradients, each against an fp64 oracle run."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import mimo_oracle as O
from tests.test_network_gpu import build_model

def run(cfg, N, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    st = O.init_state(cfg, seed)
    image = torch.rand(N, cfg.in_channels, H, W, generator=g)
    label = torch.rand(N, cfg.out_channels // 2, H, W, generator=g)
    perms = O.draw_perms(N, cfg.num_subnetworks, generator=g)
    model = build_model(cfg, st); model.train()
    out = model.training_step_with_perms(image.cuda(), label.cuda(), None, perms.cuda())
    out["loss"].backward()
    hip = {k[len("model."):]: p.grad.detach().cpu().double() for k, p in model.named_parameters()}
    res = {}
    for name, dt in (("o32", torch.float32), ("o64", torch.float64)):
        ts = O.TrainState(cfg=cfg, st={k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in st.items()},
                          loss_buffer=O.LossBuffer(cfg.num_subnetworks, 0.3, 10))
        r = O.train_step(ts, image.to(dt), label.to(dt), None, perms, apply_optimizer=False)
        res[name] = ({k: v.double() for k, v in r["grads"].items()}, r["out"].double())
    half = cfg.out_channels // 2
    preds = out["preds"].view(N, cfg.num_subnetworks, half, H, W).cpu().double()
    ref64 = res["o64"][1][:, :, :half]
    print("out: hip-vs-64 %.2e  o32-vs-64 %.2e" % (float((preds-ref64).abs().max()/ref64.abs().max()), float((res["o32"][1][:, :, :half]-ref64).abs().max()/ref64.abs().max())))
    rows = []
    for k, g64 in res["o64"][0].items():
        if k.endswith((".0.bias", ".3.bias")) and "double_conv" in k: continue
        sc = float(g64.abs().max())
        rows.append((float((hip[k]-g64).abs().max())/sc, float((res["o32"][0][k]-g64).abs().max())/sc, float((hip[k]-g64).norm()/g64.norm()), float((res["o32"][0][k]-g64).norm()/g64.norm()), k))
    rows.sort(reverse=True)
    print("%-10s %-10s %-10s %-10s name" % ("hip_max", "o32_max", "hip_rms", "o32_rms"))
    for r in rows[:14]: print("%.2e   %.2e   %.2e   %.2e   %s" % r)

run(O.NetConfig(2, 2, 2, 30), 2, 256, 256, 5)
run(O.NetConfig(3, 2, 2, 21), 2, 128, 128, 6)
run(O.NetConfig(2, 2, 4, 6), 3, 48, 64, 7)
