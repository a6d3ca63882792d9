"""Adam over the engine's flat parameter buffer: one `mimo_adam_step` launch per step
instead of one kernel chain per tensor.  Same update rule and defaults as the
``torch.optim.Adam`` the reference configures (mimo/models/mimo_unet.py:186-190)."""
from __future__ import annotations

import torch

from .engine import adam_step, adam_step_amp


class FlatAdam(torch.optim.Optimizer):
    """`torch.optim.Optimizer` (so Lightning, LR schedulers and checkpoints treat it like any
    other) whose `step()` is a single fused kernel over `net.flat_parameters()`."""

    def __init__(self, net, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        self.net = net
        super().__init__(list(net.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._step = 0
        self._m = self._v = None
        self._ref_state = None  # per-parameter torch.optim.Adam state waiting for the flat layout (load_state_dict)
        self.reduce_scale = 1.0  # multiplier applied to the gradients, e.g. 1/world_size after a sum all-reduce
        # torch.cuda.amp.GradScaler protocol for fused optimisers: the scaler sets `self.grad_scale` (tensor: divide the
        # gradients by it) and `self.found_inf` (tensor: skip the step when non-zero) around step() and deletes them
        # afterwards; unscaling, the inf / nan skip and the step counter then all happen on the device
        self._step_supports_amp_scaling = True
        self._step_dev = None

    def _init_moments(self, p: torch.Tensor) -> None:
        self._m, self._v = torch.zeros_like(p), torch.zeros_like(p)
        if self._ref_state is None:
            return
        # a reference checkpoint (torch.optim.Adam: state[i] = {step, exp_avg, exp_avg_sq} per parameter, in
        # param_groups order) scattered into the flat layout
        params = self.param_groups[0]["params"]
        base = p.data_ptr()
        for i, q in enumerate(params):
            st = self._ref_state.get(i)
            if st is None:
                continue
            off = (q.data_ptr() - base) // 4
            if off < 0 or off + q.numel() > p.numel() or st["exp_avg"].numel() != q.numel():
                raise RuntimeError(f"FlatAdam: optimiser state of parameter {i} does not fit the flat buffer")
            self._m[off: off + q.numel()].copy_(st["exp_avg"].reshape(-1))
            self._v[off: off + q.numel()].copy_(st["exp_avg_sq"].reshape(-1))
            self._step = max(self._step, int(st["step"]))
        self._ref_state = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        p, g = self.net.flat_parameters(), self.net.flat_gradients()
        if p is None:
            raise RuntimeError("FlatAdam.step() before the first forward/backward of the network")
        if self._m is None:
            self._init_moments(p)
        elif self._m.shape != p.shape:
            raise RuntimeError(f"FlatAdam: optimiser state has {self._m.numel()} elements, the network {p.numel()} — "
                               "state_dict of another architecture?")
        elif self._m.device != p.device:  # e.g. a checkpoint loaded with map_location="cpu", or module.to() since
            self._m, self._v = self._m.to(p.device), self._v.to(p.device)
        grp = self.param_groups[0]
        amp_scale, found_inf = getattr(self, "grad_scale", None), getattr(self, "found_inf", None)
        if amp_scale is not None or found_inf is not None or self._step_dev is not None:
            if self._step_dev is None or self._step_dev.device != p.device:
                self._step_dev = torch.full((1,), float(self.step_count), device=p.device, dtype=torch.float32)
            adam_step_amp(p, g, self._m, self._v, lr=float(grp["lr"]), betas=grp["betas"], eps=grp["eps"],
                          weight_decay=grp["weight_decay"], step_dev=self._step_dev, reduce_scale=self.reduce_scale,
                          amp_scale=None if amp_scale is None else amp_scale.reshape(1).float(),
                          found_inf=None if found_inf is None else found_inf.reshape(1).float())
        else:
            self._step += 1
            adam_step(p, g, self._m, self._v, lr=float(grp["lr"]), betas=grp["betas"], eps=grp["eps"],
                      weight_decay=grp["weight_decay"], step=self._step, grad_scale=self.reduce_scale)
        if hasattr(self.net, "mark_parameters_changed"):
            self.net.mark_parameters_changed()  # the kernel wrote the parameters through a raw pointer
        return loss

    @property
    def step_count(self) -> int:
        """Optimiser steps taken (skipped loss-scaler steps not counted); reads the device counter when one exists."""
        return int(self._step_dev.item()) if self._step_dev is not None else self._step

    def zero_grad(self, set_to_none: bool = True):
        # gradients are overwritten (not accumulated) by the engine when .grad is None
        super().zero_grad(set_to_none=True)

    def state_dict(self):
        d = super().state_dict()
        d["flat"] = {"step": self.step_count, "exp_avg": self._m, "exp_avg_sq": self._v}
        return d

    def load_state_dict(self, state_dict):
        """Accepts this class's own checkpoints (moments under "flat", any device) and a reference
        `torch.optim.Adam` state_dict (per-parameter `exp_avg` / `exp_avg_sq` / `step`), which is scattered into
        the flat layout at the next step.  The caller's dict is not modified."""
        sd = dict(state_dict)
        flat = sd.pop("flat", None)
        per_param = sd.get("state") or {}
        sd["state"] = {}  # the base class would try to cast per-parameter state we keep flat
        super().load_state_dict(sd)
        self._ref_state = None
        if flat is not None:
            m, v = flat["exp_avg"], flat["exp_avg_sq"]
            if (m is None) != (v is None) or (m is not None and m.shape != v.shape):
                raise ValueError("FlatAdam: inconsistent 'flat' optimiser state")
            p = self.net.flat_parameters()
            if m is not None and p is not None:
                if m.numel() != p.numel():
                    raise ValueError(f"FlatAdam: checkpoint moments have {m.numel()} elements, the network {p.numel()}")
                m, v = m.to(p.device), v.to(p.device)
            self._step = int(flat["step"])
            self._step_dev = None
            self._m = None if m is None else m.detach().clone().float()
            self._v = None if v is None else v.detach().clone().float()
        elif per_param:
            # snapshots: the source optimiser may keep stepping the tensors it handed out
            self._ref_state = {int(k): {"step": int(v["step"]), "exp_avg": v["exp_avg"].detach().clone(),
                                        "exp_avg_sq": v["exp_avg_sq"].detach().clone()} for k, v in per_param.items()}
            self._m = self._v = None
            self._step = 0
            self._step_dev = None  # a device-side step count of earlier GradScaler steps must not survive the load
