"""Does the inference / validation path block the host?  Host time of 50 un-synchronised calls against the GPU time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mimo.models.ensemble import EnsembleModule
from mimo.models.mimo_unet import MimoUnetModel

def run(name, fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); tg = time.perf_counter() - t0
    print(f"{name}: host loop {th / n * 1e3:.3f} ms/call, gpu {tg / n * 1e3:.3f} ms/call")

torch.manual_seed(0)
m = MimoUnetModel(in_channels=3, out_channels=2, num_subnetworks=1, filter_base_count=30, center_dropout_rate=0.0, final_dropout_rate=0.0,
                  encoder_dropout_rate=0.1, core_dropout_rate=0.1, decoder_dropout_rate=0.1, loss="laplace_nll", weight_decay=0.0,
                  learning_rate=1e-3, seed=0, loss_buffer_size=10, loss_buffer_temperature=0.3)
ens = EnsembleModule([], monte_carlo_steps=16, models=[m], keep_on_device=True).cuda()
for B in (1, 8):
    x = torch.randn(B, 3, 256, 256, device="cuda")
    with torch.no_grad():
        run(f"ensemble B={B}", lambda: ens(x))
m2 = MimoUnetModel(in_channels=2, out_channels=2, num_subnetworks=2, filter_base_count=30, center_dropout_rate=0.0, final_dropout_rate=0.0,
                   encoder_dropout_rate=0.0, core_dropout_rate=0.0, decoder_dropout_rate=0.0, loss="laplace_nll", weight_decay=0.0,
                   learning_rate=1e-3, seed=0, loss_buffer_size=10, loss_buffer_temperature=0.3).cuda()
m2.eval()
batch = {"image": torch.rand(16, 2, 256, 256, device="cuda"), "label": torch.rand(16, 1, 256, 256, device="cuda")}
run("validation_step N=16", lambda: m2.validation_step(batch, 0), n=20)
