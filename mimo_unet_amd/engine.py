"""Host-side handle on a `mimo_plan` (include/mimo_hip.h): geometry-specialised network
executor living in libmimo_hip.so.  PyTorch is used here only as the owner of device
memory and of the HIP stream; every computation is a C-ABI call."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L


@dataclass(frozen=True)
class TensorSpec:
    name: str
    shape: Tuple[int, ...]
    kind: int  # 0 parameter, 1 BatchNorm running buffer
    offset: int  # floats, into the flat parameter / buffer storage

    @property
    def numel(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n


@dataclass(frozen=True)
class NetGeometry:
    in_channels: int
    out_channels: int
    num_subnetworks: int
    filter_base_count: int
    encoder_dropout_rate: float = 0.0
    core_dropout_rate: float = 0.0
    decoder_dropout_rate: float = 0.0
    center_dropout_rate: float = 0.0
    final_dropout_rate: float = 0.0
    loss: str = "laplace_nll"
    # arithmetic / storage: "fp32" | "split16" | "bf16" | "bf16-mixed" | "16-mixed" (include/mimo_hip.h mimo_precision)
    precision: str = "split16"


class Plan:
    """One plan per (network geometry, batch, height, width, device)."""

    def __init__(self, geom: NetGeometry, batch: int, height: int, width: int, device: torch.device,
                 inference_only: bool = False):
        if device.type != "cuda":
            raise L.MimoHipError("the MIMO U-Net engine runs on an AMD GPU only (no CPU fallback); got " + str(device))
        self.lib = L.load()
        self.geom, self.batch, self.height, self.width, self.device = geom, batch, height, width, device
        cfg = L.MimoConfig(
            geom.in_channels, geom.out_channels, geom.num_subnetworks, geom.filter_base_count, batch, height, width,
            geom.encoder_dropout_rate, geom.core_dropout_rate, geom.decoder_dropout_rate,
            1e-5, 0.1, L.LOSS_KINDS[geom.loss], 1e-5, 1e3, device.index or 0, L.PRECISIONS[geom.precision],
            int(bool(inference_only)), geom.center_dropout_rate, geom.final_dropout_rate,
            0, 0, 0)  # BatchNorm2d + ReLU + bilinear up-sampling: the reference's blocks (the only variants that exist)
        self.inference_only = bool(inference_only)
        handle = C.c_void_p()
        with torch.cuda.device(device):
            L.check(self.lib.mimo_plan_create(C.byref(cfg), C.byref(handle)), "mimo_plan_create")
        self.handle = handle
        self.specs = self._read_specs()
        self.param_floats = int(self.lib.mimo_plan_param_floats(handle))
        self.buffer_floats = int(self.lib.mimo_plan_buffer_floats(handle))
        self.encoder_param_floats = int(self.lib.mimo_plan_encoder_param_floats(handle))
        self.backward_stages = []  # [(begin, end)] flat-gradient range that is final after each backward stage
        for st in range(int(self.lib.mimo_plan_num_backward_stages(handle))):
            b, e = C.c_int64(), C.c_int64()
            L.check(self.lib.mimo_plan_backward_stage_range(handle, st, C.byref(b), C.byref(e)), "mimo_plan_backward_stage_range")
            self.backward_stages.append((int(b.value), int(e.value)))
        self.num_double_convs = int(self.lib.mimo_plan_num_double_convs(handle))
        self.double_conv_channels = [int(self.lib.mimo_plan_double_conv_channels(handle, i))
                                     for i in range(self.num_double_convs)]
        self._bound = None
        self._ext_streams: Dict[int, torch.cuda.Stream] = {}  # the plan's own HIP streams as torch streams (backward(async_stage=True))

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            try:
                self.lib.mimo_plan_destroy(h)
            except Exception:
                pass
            self.handle = None

    def _read_specs(self) -> List[TensorSpec]:
        out = []
        n = self.lib.mimo_plan_num_tensors(self.handle)
        name = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        ndim, kind, off = C.c_int(), C.c_int(), C.c_int64()
        for i in range(n):
            L.check(self.lib.mimo_plan_tensor_info(self.handle, i, name, 256, shape, C.byref(ndim), C.byref(kind),
                                                   C.byref(off)), "mimo_plan_tensor_info")
            out.append(TensorSpec(name.value.decode(), tuple(int(shape[j]) for j in range(ndim.value)), kind.value,
                                  int(off.value)))
        return out

    @property
    def workspace_bytes(self) -> int:
        return int(self.lib.mimo_plan_workspace_bytes(self.handle))

    def bind(self, params: torch.Tensor, grads: Optional[torch.Tensor], buffers: torch.Tensor) -> None:
        key = (params.data_ptr(), 0 if grads is None else grads.data_ptr(), buffers.data_ptr())
        if key == self._bound:
            return
        assert params.numel() >= self.param_floats and buffers.numel() >= self.buffer_floats
        L.check(self.lib.mimo_plan_bind(self.handle, params.data_ptr(), L.ptr(grads), buffers.data_ptr()), "mimo_plan_bind")
        self._bound = key

    def forward(self, x: torch.Tensor, out: torch.Tensor, *, training: bool, perm: Optional[torch.Tensor] = None,
                masks: Optional[Sequence[Optional[torch.Tensor]]] = None, broadcast_subnetworks: bool = False,
                elem_masks: Optional[Sequence[Optional[torch.Tensor]]] = None, no_grad: bool = False,
                param_version: int = 0, rng: Optional[Tuple[Sequence[bool], int, int]] = None) -> None:
        """x: [N,S,Ci,H,W] (or [N,Ci,H,W] with perm / broadcast) contiguous fp32 on the plan's device.
        masks: per DoubleConv [N,C] Dropout2d multipliers; elem_masks: [center, final_0 .. final_{S-1}]
        full-shape nn.Dropout multipliers (NCHW, fp32, contiguous) or None entries.
        no_grad (eval mode only): no backward follows — BatchNorm/ReLU run in the conv epilogue.
        param_version: changes whenever parameters / BN buffers may have changed (0 = unknown); equal
        versions let eval-mode calls reuse the packed weights of the previous call.
        rng: (sites, seed, offset) — sites[i] truthy makes the engine draw the dropout multipliers of site i itself
        (DoubleConv Dropout2d sites in order, then center_dropout, then final_dropouts[s]) from a Philox stream keyed
        by (seed, offset)."""
        g = self.geom
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
        if x.dim() == 5:
            stride_n, stride_s = x.stride(0), x.stride(1)
        else:
            assert perm is not None or broadcast_subnetworks or g.num_subnetworks == 1
            stride_n, stride_s = x.stride(0), 0
        mask_arr = None
        if masks is not None and any(m is not None for m in masks):
            assert len(masks) == self.num_double_convs
            mask_arr = (C.c_void_p * self.num_double_convs)(*[L.ptr(m) or None for m in masks])
        elem_arr = None
        if elem_masks is not None and any(m is not None for m in elem_masks):
            assert len(elem_masks) == 1 + g.num_subnetworks
            for m in elem_masks:
                assert m is None or (m.is_cuda and m.dtype == torch.float32 and m.is_contiguous())
            elem_arr = (C.c_void_p * len(elem_masks))(*[L.ptr(m) or None for m in elem_masks])
        rng_arr, rng_seed, rng_offset = None, 0, 0
        if rng is not None and any(rng[0]):
            assert len(rng[0]) == self.num_double_convs + 1 + g.num_subnetworks
            rng_arr = (C.c_uint8 * len(rng[0]))(*[1 if f else 0 for f in rng[0]])
            rng_seed, rng_offset = int(rng[1]) & (2 ** 64 - 1), int(rng[2]) & (2 ** 64 - 1)
        args = L.ForwardArgs(x.data_ptr(), stride_n, stride_s, L.ptr(perm) or None, int(training),
                             C.cast(mask_arr, C.POINTER(C.c_void_p)) if mask_arr is not None else None, out.data_ptr(),
                             C.cast(elem_arr, C.POINTER(C.c_void_p)) if elem_arr is not None else None,
                             int(bool(no_grad) and not training), int(param_version), rng_arr, rng_seed, rng_offset,
                             int(x.shape[0]))
        L.check(self.lib.mimo_forward(self.handle, C.byref(args), L.current_stream()), "mimo_forward")

    STATUS_BITS = {1: "a BatchNorm batch statistic of a training forward was not finite",
                   2: "a BatchNorm-backward sum was not finite",
                   4: "a logit was not finite"}

    def status(self, clear: bool = True) -> int:
        """Numerics status word of the plan's kernels since the last clear (mimo_plan_status; synchronises the
        current stream).  0 = nothing recorded; bits: STATUS_BITS."""
        flags = C.c_int32(0)
        L.check(self.lib.mimo_plan_status(self.handle, C.byref(flags), 1 if clear else 0, L.current_stream()), "mimo_plan_status")
        return int(flags.value)

    def dropout_mask(self, site: int) -> torch.Tensor:
        """Dropout multipliers of the last forward: site < num_double_convs -> [N, C]; num_double_convs + j -> the
        element-wise site j (0 center, 1 + s final s) as [N, C, H', W'] (only when the engine drew it)."""
        g = self.geom
        S, f, n = g.num_subnetworks, g.filter_base_count, self.batch
        if site < self.num_double_convs:
            shape = (n, self.double_conv_channels[site])
        elif site == self.num_double_convs:
            shape = (n, 8 * f * S, self.height // 16, self.width // 16)
        else:
            shape = (n, f, self.height, self.width)
        out = torch.empty(shape, device=self.device, dtype=torch.float32)
        L.check(self.lib.mimo_plan_dropout_mask(self.handle, site, out.data_ptr(), L.current_stream()), "mimo_plan_dropout_mask")
        return out

    def loss_forward(self, label: torch.Tensor, mask: Optional[torch.Tensor], perm: Optional[torch.Tensor],
                     loss_out: torch.Tensor) -> None:
        L.check(self.lib.mimo_loss_forward(self.handle, label.data_ptr(), L.ptr(mask) or None, L.ptr(perm) or None,
                                           loss_out.data_ptr(), L.current_stream()), "mimo_loss_forward")

    def backward(self, dout: Optional[torch.Tensor], dloss: Optional[torch.Tensor], dx: Optional[torch.Tensor],
                 stage: Optional[int] = None, async_stage: bool = False):
        """stage None: whole backward; else one of the `backward_stages`, in order (include/mimo_hip.h).
        async_stage (with a stage): the current stream is not made to wait for the plan's side stream; returns the
        torch stream on which the stage's gradient range is final (None: the current stream) — the data-parallel caller
        issues that range's collective there (mimo_backward_stage_async)."""
        if stage is None:
            L.check(self.lib.mimo_backward(self.handle, L.ptr(dout) or None, L.ptr(dloss) or None, L.ptr(dx) or None,
                                           L.current_stream()), "mimo_backward")
        elif not async_stage:
            L.check(self.lib.mimo_backward_stage(self.handle, stage, L.ptr(dout) or None, L.ptr(dloss) or None,
                                                 L.ptr(dx) or None, L.current_stream()), "mimo_backward_stage")
        else:
            cur = L.current_stream()
            ready = C.c_void_p()
            L.check(self.lib.mimo_backward_stage_async(self.handle, stage, L.ptr(dout) or None, L.ptr(dloss) or None,
                                                       L.ptr(dx) or None, cur, C.byref(ready)), "mimo_backward_stage_async")
            if ready.value is not None and ready.value != cur:
                ext = self._ext_streams.get(ready.value)
                if ext is None:
                    ext = self._ext_streams[ready.value] = torch.cuda.ExternalStream(ready.value, device=self.device)
                return ext
        return None


    PROF_KINDS = ("conv3x3_fwd", "conv3x3_dgrad", "conv3x3_wgrad", "bn_relu_fwd", "bn_bwd_reduce", "bn_bwd_apply",
                  "upcat_fwd", "up_bwd", "pool_bwd", "head_fwd", "head_bwd")
    PROF_TIERS = 5  # resolution levels: 0 = H x W ... 4 = H/16 x W/16

    def profile(self, enable: bool) -> None:
        L.check(self.lib.mimo_plan_profile(self.handle, int(enable)), "mimo_plan_profile")

    def profile_read(self) -> Dict[str, Dict[str, float]]:
        res = {}
        for k, name in enumerate(self.PROF_KINDS):
            ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
            L.check(self.lib.mimo_plan_profile_read(self.handle, k, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)),
                    "mimo_plan_profile_read")
            res[name] = {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}
        return res

    def profile_read_kind_tiers(self) -> Dict[str, List[float]]:
        """{kernel class: [ms per resolution tier]} — which kernels a tier's device time is made of."""
        res = {}
        for k, name in enumerate(self.PROF_KINDS):
            row = []
            for t in range(self.PROF_TIERS):
                ms = C.c_double()
                L.check(self.lib.mimo_plan_profile_read_kind_tier(self.handle, k, t, C.byref(ms)), "mimo_plan_profile_read_kind_tier")
                row.append(ms.value)
            res[name] = row
        return res

    def profile_read_tiers(self) -> List[Tuple[float, float]]:
        """[(forward_ms, backward_ms)] per resolution tier: device time of every launch of the tier's blocks."""
        res = []
        for t in range(self.PROF_TIERS):
            f, b = C.c_double(), C.c_double()
            L.check(self.lib.mimo_plan_profile_read_tier(self.handle, t, C.byref(f), C.byref(b)), "mimo_plan_profile_read_tier")
            res.append((f.value, b.value))
        return res


def adam_step(params: torch.Tensor, grads: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, *, lr: float,
              betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, step: int, grad_scale: float = 1.0) -> None:
    lib = L.load()
    L.check(lib.mimo_adam_step(params.data_ptr(), grads.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                               params.numel(), lr, betas[0], betas[1], eps, weight_decay, step, grad_scale,
                               L.current_stream()), "mimo_adam_step")


def adam_step_amp(params: torch.Tensor, grads: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, *, lr: float,
                  betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, step_dev: torch.Tensor,
                  reduce_scale: float = 1.0, amp_scale: Optional[torch.Tensor] = None,
                  found_inf: Optional[torch.Tensor] = None) -> None:
    """Adam under a loss scaler: device-side step counter, gradients divided by `amp_scale`, the whole update skipped
    when `found_inf` is non-zero (include/mimo_hip.h mimo_adam_step_amp)."""
    lib = L.load()
    for t in (step_dev, amp_scale, found_inf):
        assert t is None or (t.is_cuda and t.dtype == torch.float32 and t.numel() == 1)
    L.check(lib.mimo_adam_step_amp(params.data_ptr(), grads.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                   params.numel(), lr, betas[0], betas[1], eps, weight_decay, step_dev.data_ptr(),
                                   reduce_scale, L.ptr(amp_scale) or None, L.ptr(found_inf) or None, L.current_stream()),
            "mimo_adam_step_amp")


def uncertainties(p1: torch.Tensor, p2: torch.Tensor, loss: str = "laplace_nll"):
    """[N,S,C,H,W] x2 -> (mean, aleatoric_var, epistemic_var) each [N,C,H,W], on device."""
    lib = L.load()
    p1, p2 = p1.contiguous(), p2.contiguous()
    n, s, c, h, w = p1.shape
    outs = [torch.empty(n, c, h, w, device=p1.device, dtype=torch.float32) for _ in range(3)]
    L.check(lib.mimo_uncertainties(p1.data_ptr(), p2.data_ptr(), n, s, c, h * w, L.LOSS_KINDS[loss], outs[0].data_ptr(),
                                   outs[1].data_ptr(), outs[2].data_ptr(), L.current_stream()), "mimo_uncertainties")
    return tuple(outs)


class _EvidentialHeadLoss(torch.autograd.Function):
    """(ev, loss_map) = evidential head + loss of the backbone logits [B,4,H,W] in one kernel; the backward is one
    kernel too (include/mimo_hip.h mimo_evidential_forward / _backward)."""

    @staticmethod
    def forward(ctx, logits, label, mask):
        lib = L.load()
        logits = logits.contiguous().float()
        b, four, h, w = logits.shape
        assert four == 4 and logits.is_cuda
        label_c = None if label is None else label.reshape(b, h * w).contiguous().float()
        mask_c = None if mask is None else mask.reshape(b, h * w).contiguous().float()
        ev = torch.empty_like(logits)
        loss = torch.empty(b, h, w, device=logits.device, dtype=torch.float32) if label is not None else None
        L.check(lib.mimo_evidential_forward(logits.data_ptr(), L.ptr(label_c) or None, L.ptr(mask_c) or None, b, h * w,
                                            ev.data_ptr(), L.ptr(loss) or None, L.current_stream()), "mimo_evidential_forward")
        ctx.save_for_backward(logits, label_c, mask_c)
        if loss is None:
            loss = torch.zeros(0, device=logits.device)
            ctx.mark_non_differentiable(loss)
        return ev, loss

    @staticmethod
    def backward(ctx, d_ev, d_loss):
        lib = L.load()
        logits, label_c, mask_c = ctx.saved_tensors
        b, _, h, w = logits.shape
        d_ev = None if d_ev is None else d_ev.contiguous().float()
        d_loss = None if (d_loss is None or label_c is None) else d_loss.contiguous().float()
        out = torch.empty_like(logits)
        L.check(lib.mimo_evidential_backward(logits.data_ptr(), L.ptr(label_c) or None, L.ptr(mask_c) or None,
                                             L.ptr(d_ev) or None, L.ptr(d_loss) or None, b, h * w, out.data_ptr(),
                                             L.current_stream()), "mimo_evidential_backward")
        return out, None, None


def evidential_head_loss(logits: torch.Tensor, label: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None):
    """Backbone logits [B,4,H,W] (+ label [B,1,H,W], mask [B,H,W]) -> (ev [B,4,H,W] = (gamma, v, alpha, beta),
    per-pixel loss [B,H,W] or None), differentiable w.r.t. the logits."""
    ev, loss = _EvidentialHeadLoss.apply(logits, label, mask)
    return ev, (loss if label is not None else None)


class _LossBufferStep(torch.autograd.Function):
    """(mean(loss * weights), weights, mean(loss)) with `loss` written into row `index` of the ring — one launch
    (include/mimo_hip.h mimo_loss_buffer_step).  d mean(loss * weights) / d loss = weights / S: the weights carry no
    gradient, they come from the DETACHED losses of earlier steps (mimo_unet.py:243-245)."""

    @staticmethod
    def forward(ctx, loss, ring, index, temperature):
        lib = L.load()
        size, S = ring.shape
        ld = loss.detach()
        if not (ld.is_contiguous() and ld.dtype == torch.float32):
            ld = ld.contiguous().float()
        out = torch.empty(2 * S + 2, device=ring.device, dtype=torch.float32)
        weights, w_over_s, scalars = out[:S], out[S:2 * S], out[2 * S:]
        L.check(lib.mimo_loss_buffer_step(ring.data_ptr(), size, index, S, float(temperature), ld.data_ptr(), weights.data_ptr(),
                                          w_over_s.data_ptr(), scalars.data_ptr(), L.current_stream()), "mimo_loss_buffer_step")
        ctx.save_for_backward(w_over_s)
        train_loss = scalars[1]
        ctx.mark_non_differentiable(weights, train_loss)
        return scalars[0], weights, train_loss

    @staticmethod
    def backward(ctx, grad, _gw, _gt):
        (w_over_s,) = ctx.saved_tensors
        return grad * w_over_s, None, None, None


def loss_buffer_step(ring: torch.Tensor, index: int, temperature: float, loss: torch.Tensor):
    """One launch for MimoUnetModel._calculate_train_loss's [S] arithmetic: returns (the differentiable mean(loss * weights),
    weights [S] read before `loss` enters the ring, mean(loss)); row `index` of `ring` [size, S] now holds the detached loss."""
    assert ring.is_cuda and ring.dtype == torch.float32 and ring.is_contiguous() and loss.shape == (ring.shape[1],)
    return _LossBufferStep.apply(loss, ring, index, temperature)


VAL_SCALARS = ("nll_combined", "mae", "mse", "rmse", "r2", "aleatoric_std_mean", "epistemic_std_mean", "count")


TRAIN_SCALARS = ("mae", "mse", "rmse", "r2", "count")


def training_epilogue(out: torch.Tensor, label: torch.Tensor, perms: Optional[torch.Tensor], loss: str = "laplace_nll"):
    """Logits [N,S,2*Ct,H,W] + label [N0,Ct,H,W] + perms [S,N] -> (label_t, preds, aleatoric_std, err) as
    [N,S,Ct,H,W] and a device tensor of the TRAIN_SCALARS — the no_grad tail of MimoUnetModel.training_step
    (mimo_unet.py:121-144: label gather, mode / std, error map, regression metrics) in one pass."""
    lib = L.load()
    out, label = out.detach().contiguous(), label.contiguous().float()
    n, s, co, h, w = out.shape
    ct = co // 2
    if perms is not None:
        perms = perms.to(device=out.device, dtype=torch.int64).contiguous()
        assert perms.shape == (s, n), (perms.shape, out.shape)
    else:
        assert label.shape[0] == n
    assert label.shape[1:] == (ct, h, w), (label.shape, out.shape)
    maps = [torch.empty(n, s, ct, h, w, device=out.device, dtype=torch.float32) for _ in range(4)]
    scalars = torch.empty(5, device=out.device, dtype=torch.float32)
    blocks = 2048
    scratch = torch.empty(blocks * 8, device=out.device, dtype=torch.float64)
    L.check(lib.mimo_training_epilogue(out.data_ptr(), label.data_ptr(), L.ptr(perms) or None, n, s, ct, h * w,
                                       L.LOSS_KINDS[loss], maps[0].data_ptr(), maps[1].data_ptr(), maps[2].data_ptr(),
                                       maps[3].data_ptr(), scalars.data_ptr(), scratch.data_ptr(), blocks,
                                       L.current_stream()), "mimo_training_epilogue")
    return maps[0], maps[1], maps[2], maps[3], scalars


def validation_epilogue(out: torch.Tensor, label: torch.Tensor, mask: Optional[torch.Tensor], loss: str = "laplace_nll",
                        eps_min: float = 1e-5, eps_max: float = 1e3):
    """Logits [N,S,2*Ct,H,W] + label [N,Ct,H,W] (+ mask [N,1,H,W]) -> (mean, aleatoric_std, epistemic_std,
    err) maps [N,Ct,H,W] and a device tensor of the eight scalars named in VAL_SCALARS — the tail of
    MimoUnetModel.validation_step (mimo_unet.py:153-183) in one pass."""
    lib = L.load()
    out, label = out.contiguous(), label.contiguous().float()
    n, s, co, h, w = out.shape
    ct = co // 2
    assert label.shape == (n, ct, h, w), (label.shape, out.shape)
    mask = None if mask is None else mask.contiguous().float()
    maps = [torch.empty(n, ct, h, w, device=out.device, dtype=torch.float32) for _ in range(4)]
    scalars = torch.empty(8, device=out.device, dtype=torch.float32)
    blocks = 1024
    scratch = torch.empty(blocks * 8, device=out.device, dtype=torch.float64)
    L.check(lib.mimo_validation_epilogue(out.data_ptr(), label.data_ptr(), L.ptr(mask) or None, n, s, ct, h * w,
                                         L.LOSS_KINDS[loss], eps_min, eps_max, maps[0].data_ptr(), maps[1].data_ptr(),
                                         maps[2].data_ptr(), maps[3].data_ptr(), scalars.data_ptr(), scratch.data_ptr(),
                                         blocks, L.current_stream()), "mimo_validation_epilogue")
    return maps[0], maps[1], maps[2], maps[3], scalars
