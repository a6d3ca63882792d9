"""`EnsembleModule`: deep-ensemble / MC-dropout inference with the reference's interface
(``mimo/models/ensemble.py:9-115``), kept on the GPU.

Same constructor (`checkpoint_paths, monte_carlo_steps=0, return_raw_predictions=False`),
same `num_subnetworks` / `loss_fn` properties, same return values.  Differences underneath:
the Monte-Carlo passes of one checkpoint run as ONE batched launch sequence (the passes are
stacked on the batch axis, each sample draws its own Dropout2d mask, BatchNorm is in eval
mode so samples do not interact), predictions never bounce through host memory between
passes, and the subnetwork-axis reduction is the `mimo_uncertainties` kernel.  Results are
returned on the host like the reference does (it moves every pass with `.cpu()`,
ensemble.py:101-102) unless `keep_on_device=True`."""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch

from ..lightning_compat import LightningModule
from .mimo_unet import MimoUnetModel
from .utils import compute_uncertainties


class EnsembleModule(LightningModule):
    def __init__(self, checkpoint_paths: List[str], monte_carlo_steps: int = 0, return_raw_predictions: bool = False,
                 models: Optional[Sequence[MimoUnetModel]] = None, keep_on_device: bool = False):
        super().__init__()
        self.models = list(models) if models is not None else [MimoUnetModel.load_from_checkpoint(p) for p in checkpoint_paths]
        self.monte_carlo_steps = monte_carlo_steps
        self.return_raw_predictions = return_raw_predictions
        self.keep_on_device = keep_on_device
        self.max_samples_per_launch = max(1, int(os.environ.get("MIMO_MC_CHUNK_SAMPLES", "64")))
        self._run_device = None
        for model in self.models:
            model.eval()
            if self.monte_carlo_steps > 0:
                self._activate_mc_dropout(model)

    @staticmethod
    def _activate_mc_dropout(model: torch.nn.Module):
        """Put every `Dropout*` child back in training mode (BatchNorm stays in eval)."""
        for submodule in model.modules():
            if submodule.__class__.__name__.startswith("Dropout"):
                submodule.train()

    @property
    def num_subnetworks(self) -> int:
        return sum(model.num_subnetworks for model in self.models)

    @property
    def loss_fn(self):
        return self.models[0].loss_fn

    # models are held in a plain list (as in the reference), so `.to()` must be forwarded by hand
    def _apply(self, fn, *a, **k):
        for m in self.models:
            m._apply(fn, *a, **k)
        return super()._apply(fn, *a, **k)

    @property
    def device(self) -> torch.device:
        for m in self.models:
            for p in m.parameters():
                return p.device
        return torch.device("cpu")

    def forward(self, x: torch.Tensor):
        """x [B,C_in,H,W] -> (mean, aleatoric_variance, epistemic_variance) [B,C_out,H,W], or the raw
        (p1, p2) [B,S_total,C_out,H,W] when `return_raw_predictions`."""
        passes = max(1, self.monte_carlo_steps)
        b = x.shape[0]
        p1_list, p2_list = [], []
        with torch.no_grad():
            for model in self.models:
                dev = next(model.parameters()).device
                xd = x.to(dev)
                s, co = model.num_subnetworks, model.out_channels
                # passes are stacked on the batch axis in chunks of at most `max_samples_per_launch` samples
                # (the reference loops over passes in O(B) memory; one launch sequence over passes x B samples
                # would need a passes-times larger plan)
                chunk = max(1, min(passes, self.max_samples_per_launch // max(b, 1)))
                net = model.model
                recorded = net.mask_override  # tests: recorded Dropout2d masks for all passes x B samples
                outs = []
                try:
                    for m0 in range(0, passes, chunk):
                        m1 = min(passes, m0 + chunk)
                        xb = xd if m1 - m0 == 1 else xd.repeat(m1 - m0, 1, 1, 1)  # pass-major: (m, i) at row (m-m0)*B + i
                        n = xb.shape[0]
                        ident = torch.arange(n, device=dev, dtype=torch.int64)[None].repeat(s, 1)
                        if recorded is not None:
                            net.mask_override = {j: t[m0 * b: m1 * b] for j, t in recorded.items()}
                        outs.append(net._call(xb, None, None, ident)[0])
                finally:
                    net.mask_override = recorded
                out = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
                # [passes*B, S, Co, H, W] -> [B, passes*S, Co, H, W] (pass-major on the subnetwork axis, like
                # the reference's torch.cat of per-pass outputs)
                out = out.view(passes, b, s, co, *out.shape[-2:]).permute(1, 0, 2, 3, 4, 5)
                out = out.reshape(b, passes * s, co, *out.shape[-2:])
                p1_list.append(out[:, :, : co // 2])
                p2_list.append(out[:, :, co // 2:])
            p1 = torch.cat(p1_list, dim=1).contiguous()
            p2 = torch.cat(p2_list, dim=1).contiguous()
            if self.return_raw_predictions:
                res = (p1, p2)
            else:
                res = compute_uncertainties(self.loss_fn, y_preds=p1, log_params=p2)
        return res if self.keep_on_device else tuple(t.cpu() for t in res)
