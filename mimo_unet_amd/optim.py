"""Adam over the engine's flat parameter buffer: one `mimo_adam_step` launch per step
instead of one kernel chain per tensor.  Same update rule and defaults as the
``torch.optim.Adam`` the reference configures (mimo/models/mimo_unet.py:186-190)."""
from __future__ import annotations

import torch

from .engine import adam_step


class FlatAdam(torch.optim.Optimizer):
    """`torch.optim.Optimizer` (so Lightning, LR schedulers and checkpoints treat it like any
    other) whose `step()` is a single fused kernel over `net.flat_parameters()`."""

    def __init__(self, net, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 extra_params=()):
        self.net = net
        params = list(net.parameters()) + list(extra_params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._step = 0
        self._m = self._v = None
        self.grad_scale = 1.0  # e.g. 1/world_size after a sum all-reduce

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        p, g = self.net.flat_parameters(), self.net.flat_gradients()
        if p is None:
            raise RuntimeError("FlatAdam.step() before the first forward/backward of the network")
        if self._m is None or self._m.data_ptr() == 0 or self._m.shape != p.shape or self._m.device != p.device:
            self._m, self._v = torch.zeros_like(p), torch.zeros_like(p)
        grp = self.param_groups[0]
        self._step += 1
        adam_step(p, g, self._m, self._v, lr=float(grp["lr"]), betas=grp["betas"], eps=grp["eps"],
                  weight_decay=grp["weight_decay"], step=self._step, grad_scale=self.grad_scale)
        if hasattr(self.net, "mark_parameters_changed"):
            self.net.mark_parameters_changed()  # the kernel wrote the parameters through a raw pointer
        return loss

    def zero_grad(self, set_to_none: bool = True):
        # gradients are overwritten (not accumulated) by the engine when .grad is None
        super().zero_grad(set_to_none=True)

    def state_dict(self):
        d = super().state_dict()
        d["flat"] = {"step": self._step, "exp_avg": self._m, "exp_avg_sq": self._v}
        return d

    def load_state_dict(self, state_dict):
        flat = state_dict.pop("flat", None)
        super().load_state_dict(state_dict)
        if flat is not None:
            self._step, self._m, self._v = flat["step"], flat["exp_avg"], flat["exp_avg_sq"]
