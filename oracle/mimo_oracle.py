"""CPU oracle for the MIMO U-Net train/inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``mimo_unet_amd/`` may import this
module: it is the checker for ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  The product path is the HIP library.

This is a *functional* restatement (plain dict of tensors in, tensors out, no
nn.Module tree) of what the reference computes, written from SURVEY.md §8(a).
The leaf arithmetic of the reference lives in a third-party dependency that is
present here and on the GPU box (PyTorch, ``torch==2.*`` in the reference's
requirements.txt:2; this image: 2.10.0), so the oracle calls the same
``torch.nn.functional`` leaf ops the reference's modules dispatch to, on CPU,
fp32.  Parity pinning: ``tests/golden/*.npz`` are produced by
``tests/golden/make_golden.py`` from the *imported reference itself*
(``/root/reference``) and ``tests/test_oracle_golden.py`` checks this file
against them, so the oracle is pinned to reference outputs, not to itself.

Reference citations (relative to /root/reference):
  DoubleConv / Down / Up / OutConv ... mimo/models/mimo_components/components.py:8-129
  MimoUNet topology ................. mimo/models/mimo_components/model.py:26-297
  LaplaceNLL / GaussianNLL .......... mimo/losses.py:39-192
  LossBuffer ........................ mimo/models/mimo_components/loss_buffer.py:3-74
  input transform / uncertainties ... mimo/models/utils.py:5-101
  train step / optimiser ............ mimo/models/mimo_unet.py:115-144,185-201,223-247
"""
from __future__ import annotations

import contextlib
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------
# configuration + parameter naming (state_dict names of the reference)
# --------------------------------------------------------------------------
@dataclass
class NetConfig:
    in_channels: int
    out_channels: int  # total head width = targets * 2  (mimo_unet.py:110-111)
    num_subnetworks: int
    filter_base_count: int
    encoder_dropout_rate: float = 0.0
    core_dropout_rate: float = 0.0
    decoder_dropout_rate: float = 0.0
    center_dropout_rate: float = 0.0
    final_dropout_rate: float = 0.0


def double_conv_specs(cfg: NetConfig) -> List[Tuple[str, int, int, int]]:
    """(state_dict prefix, Cin, Cmid, Cout) of every DoubleConv, in forward order.

    Widths follow model.py:119-297 with bilinear=True (factor 2) as hard-wired by
    mimo_unet.py:73-74; ``Up`` uses mid = in // 2 (components.py:79-85).
    """
    f, S = cfg.filter_base_count, cfg.num_subnetworks
    specs = []
    for s in range(S):
        specs.append((f"encoder.in_convs.{s}.double_conv", cfg.in_channels, f, f))
    for s in range(S):
        specs.append((f"encoder.down1s.{s}.conv.double_conv", f, 2 * f, 2 * f))
    specs.append(("core.down2.conv.double_conv", 2 * f * S, 4 * f * S, 4 * f * S))
    specs.append(("core.down3.conv.double_conv", 4 * f * S, 8 * f * S, 8 * f * S))
    specs.append(("core.down4.conv.double_conv", 8 * f * S, 8 * f * S, 8 * f * S))
    specs.append(("core.up1.conv.double_conv", 16 * f * S, 8 * f * S, 4 * f * S))
    specs.append(("core.up2.conv.double_conv", 8 * f * S, 4 * f * S, 2 * f * S))
    specs.append(("core.up3.conv.double_conv", 4 * f * S, 2 * f * S, f * S))
    cin = f * S + f
    for s in range(S):
        specs.append((f"decoder.up4s.{s}.conv.double_conv", cin, cin // 2, f))
    return specs


def param_shapes(cfg: NetConfig) -> Dict[str, Tuple[int, ...]]:
    """Every state_dict entry (parameters and BN buffers) with its shape."""
    shapes: Dict[str, Tuple[int, ...]] = {}
    for prefix, cin, cmid, cout in double_conv_specs(cfg):
        for conv_i, bn_i, ci, co in ((0, 1, cin, cmid), (3, 4, cmid, cout)):
            shapes[f"{prefix}.{conv_i}.weight"] = (co, ci, 3, 3)
            shapes[f"{prefix}.{conv_i}.bias"] = (co,)
            shapes[f"{prefix}.{bn_i}.weight"] = (co,)
            shapes[f"{prefix}.{bn_i}.bias"] = (co,)
            shapes[f"{prefix}.{bn_i}.running_mean"] = (co,)
            shapes[f"{prefix}.{bn_i}.running_var"] = (co,)
            shapes[f"{prefix}.{bn_i}.num_batches_tracked"] = ()
    for s in range(cfg.num_subnetworks):
        shapes[f"decoder.outcs.{s}.conv.weight"] = (cfg.out_channels, cfg.filter_base_count, 1, 1)
        shapes[f"decoder.outcs.{s}.conv.bias"] = (cfg.out_channels,)
    return shapes


def is_buffer(name: str) -> bool:
    return name.endswith(("running_mean", "running_var", "num_batches_tracked"))


def init_state(cfg: NetConfig, seed: int) -> Dict[str, Tensor]:
    """Random state with PyTorch's default init *distributions* (kaiming-uniform
    a=sqrt(5) conv weights, uniform(+-1/sqrt(fan_in)) bias, BN gamma=1 beta=0).
    Values are not bit-equal to ``torch.nn.Conv2d`` under the same seed (draw
    order differs); goldens carry the reference's actual state instead."""
    g = torch.Generator().manual_seed(seed)
    state: Dict[str, Tensor] = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith("num_batches_tracked"):
            state[name] = torch.zeros((), dtype=torch.int64)
        elif name.endswith("running_mean"):
            state[name] = torch.zeros(shape)
        elif name.endswith("running_var"):
            state[name] = torch.ones(shape)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            bound = 1.0 / math.sqrt(fan_in)  # kaiming_uniform(a=sqrt5) == U(+-1/sqrt(fan_in))
            state[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif ".double_conv.1." in name or ".double_conv.4." in name:
            state[name] = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
        else:  # conv bias
            wshape = param_shapes(cfg)[name[: -len("bias")] + "weight"]
            bound = 1.0 / math.sqrt(wshape[1] * wshape[2] * wshape[3])
            state[name] = (torch.rand(shape, generator=g) * 2 - 1) * bound
    return state


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------
_CONV_OPERANDS = "fp32"
_STORE16 = None  # None | torch.bfloat16 | torch.float16: 16-bit STORAGE of conv outputs and activations
_STORE_GRADS = False  # ... and of their gradients (da, dz, the padded-domain data gradient) in the backward


class conv_operands:
    """Context manager: ``with conv_operands("bf16"):`` makes every 3x3 convolution round its operands to
    bf16 — input and weight in the forward, the upstream gradient too in the data / weight gradients —
    with fp32 accumulation and an fp32 bias: the arithmetic of the engine's MIMO_PREC_BF16 mode (what
    autocast does to a conv in the reference's ``precision="16-mixed"`` runs, scripts/train/train_ndvi.py:71).
    No reference fixture pins this mode: it is the pinned fp32 restatement with roundings inserted.

    ``"bf16-mixed"`` / ``"16-mixed"`` (the engine's MIMO_PREC_BF16_MIXED / MIMO_PREC_FP16_MIXED, i.e. Lightning's
    autocast precisions): operands rounded to bf16 / fp16 in every convolution but the image convolution, and the
    FORWARD storage roundings of the engine inserted — convolution outputs and activations are stored in that type
    (BatchNorm statistics come from the unrounded fp32 accumulators; the 1x1 head reads the stored activation and
    produces fp32 logits).  By default the backward of this emulation keeps fp32 gradients; ``grad_storage=True``
    also rounds the gradients where the engine stores them in 16 bits — the gradient of every stored activation (da),
    of every stored convolution output (dz) and the data gradient of a convolution on the padded domain — which is
    what the per-tensor gradient checks of the storage modes anchor on (round 3; under "16-mixed" run it with the same
    loss scale as the engine: fp16 rounding is not scale-invariant near the subnormals)."""

    def __init__(self, kind: str, grad_storage: bool = False):
        assert kind in ("fp32", "bf16", "bf16-mixed", "16-mixed")
        self.kind = kind
        self.grad_storage = grad_storage

    def __enter__(self):
        global _CONV_OPERANDS, _STORE16, _STORE_GRADS
        self.prev = (_CONV_OPERANDS, _STORE16, _STORE_GRADS)
        _CONV_OPERANDS = self.kind
        _STORE16 = {"bf16-mixed": torch.bfloat16, "16-mixed": torch.float16}.get(self.kind)
        _STORE_GRADS = self.grad_storage and _STORE16 is not None

    def __exit__(self, *exc):
        global _CONV_OPERANDS, _STORE16, _STORE_GRADS
        _CONV_OPERANDS, _STORE16, _STORE_GRADS = self.prev


class _StoreRound(torch.autograd.Function):
    """Round to the 16-bit storage type in the forward; in the backward identity, or (conv_operands(grad_storage=True))
    the same rounding of the gradient: the engine stores the gradient of a stored tensor in the same type."""

    @staticmethod
    def forward(ctx, t, dtype):
        ctx.dtype, ctx.round_grad = dtype, _STORE_GRADS
        return t.to(dtype).float()

    @staticmethod
    def backward(ctx, g):
        return (g.to(ctx.dtype).float() if ctx.round_grad else g), None


def _stored(t: Tensor) -> Tensor:
    return t if _STORE16 is None else _StoreRound.apply(t, _STORE16)


class _RoundedConv(torch.autograd.Function):
    """The engine's MIMO_PREC_BF16 policy per convolution (plan.hip init_convbn): the forward runs on the
    bf16 MFMA when the padded input has >= 16 channels (C_in > 8), the data gradient when the padded output
    has >= 16 channels (C_out > 8), the weight gradient always; the other cases stay exact fp32."""

    @staticmethod
    def forward(ctx, xp, w):
        dt = _STORE16 or torch.bfloat16
        r = lambda t: t.to(dt).float()
        ctx.save_for_backward(xp, w)
        ctx.dt, ctx.mixed, ctx.round_grad = dt, _STORE16 is not None, _STORE_GRADS
        # MIMO_PREC_BF16: 16-bit forward when the padded input has >= 16 channels; mixed storage modes: every layer
        # but the image convolution (C_in <= 4)
        low = w.shape[1] > 4 if ctx.mixed else w.shape[1] > 8
        return F.conv2d(r(xp), r(w)) if low else F.conv2d(xp, w)

    @staticmethod
    def backward(ctx, dz):
        r = lambda t: t.to(ctx.dt).float()
        xp, w = ctx.saved_tensors
        low = ctx.mixed or w.shape[0] > 8
        dx = torch.nn.grad.conv2d_input(xp.shape, r(w), r(dz)) if low else torch.nn.grad.conv2d_input(xp.shape, w, dz)
        if ctx.round_grad and low:
            dx = r(dx)  # the padded-domain data gradient is a 16-bit tensor in HBM
        xin = r(xp) if (not ctx.mixed or w.shape[1] > 4) else xp.to(ctx.dt).float()  # the fp32 image is rounded on load
        return dx, torch.nn.grad.conv2d_weight(xin, w.shape, r(dz))


def conv3x3_reflect(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """Conv2d(k=3, padding=1, padding_mode='reflect') — components.py:23,26."""
    xp = F.pad(x, (1, 1, 1, 1), mode="reflect")
    if _CONV_OPERANDS != "fp32":
        z = _RoundedConv.apply(xp, w)
        return z if b is None else z + b[None, :, None, None]
    return F.conv2d(xp, w, b)


def conv_bn_relu(x: Tensor, st: Dict[str, Tensor], conv: str, bn: str, training: bool) -> Tensor:
    """conv3x3(reflect)+BatchNorm2d+ReLU — components.py:23-25 / :26-28.
    Training mode normalises with biased batch variance and updates the running
    buffers in ``st`` in place (momentum 0.1, unbiased variance)."""
    z = conv3x3_reflect(x, st[conv + ".weight"], st[conv + ".bias"])
    if _STORE16 is not None and x.shape[1] > 4 and not training:
        # 16-bit storage of the conv output (eval mode: the normalisation constants do not depend on z; in training
        # mode the engine takes the statistics from the fp32 accumulators and normalises the STORED z — emulated
        # below by normalising the rounded tensor with the statistics of the unrounded one)
        z = _stored(z)
    if _STORE16 is not None and x.shape[1] > 4 and training:
        mean = z.mean(dim=(0, 2, 3))
        var = z.var(dim=(0, 2, 3), unbiased=False)
        n = z.numel() // z.shape[1]
        with torch.no_grad():
            st[bn + ".running_mean"] = (1 - BN_MOMENTUM) * st[bn + ".running_mean"] + BN_MOMENTUM * mean
            st[bn + ".running_var"] = (1 - BN_MOMENTUM) * st[bn + ".running_var"] + BN_MOMENTUM * var * n / max(n - 1, 1)
        zs = _stored(z)
        z = (zs - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS)
        z = z * st[bn + ".weight"][None, :, None, None] + st[bn + ".bias"][None, :, None, None]
    else:
        z = F.batch_norm(
            z, st[bn + ".running_mean"], st[bn + ".running_var"], st[bn + ".weight"], st[bn + ".bias"],
            training, BN_MOMENTUM, BN_EPS,
        )
    if training and (bn + ".num_batches_tracked") in st:
        st[bn + ".num_batches_tracked"] += 1
    return _stored(F.relu(z))


def channel_dropout(x: Tensor, p: float, active: bool, mask: Optional[Tensor]) -> Tensor:
    """Dropout2d (components.py:29).  ``mask`` [N, C] holds the multipliers
    (0 or 1/(1-p)) so that tests can inject the same draw into the HIP path."""
    if not active or p == 0.0:
        return x
    if mask is None:
        return F.dropout2d(x, p, True)
    return x * mask[:, :, None, None].to(x.dtype)


def double_conv(x: Tensor, st: Dict[str, Tensor], prefix: str, training: bool,
                drop_p: float = 0.0, drop_active: bool = False, drop_mask: Optional[Tensor] = None) -> Tensor:
    """DoubleConv.forward — components.py:8-33."""
    h = conv_bn_relu(x, st, prefix + ".0", prefix + ".1", training)
    h = conv_bn_relu(h, st, prefix + ".3", prefix + ".4", training)
    return channel_dropout(h, drop_p, drop_active, drop_mask)


def up_cat(low: Tensor, skip: Tensor) -> Tensor:
    """Up.forward up to the concat — components.py:106-119: bilinear x2
    (align_corners=True), zero-pad to the skip's H, W, cat([skip, up])."""
    up = F.interpolate(low, scale_factor=2, mode="bilinear", align_corners=True)
    dy = skip.shape[2] - up.shape[2]
    dx = skip.shape[3] - up.shape[3]
    up = F.pad(up, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    return torch.cat([skip, _stored(up)], dim=1)


# --------------------------------------------------------------------------
# network forward
# --------------------------------------------------------------------------
def mimo_unet_forward(cfg: NetConfig, st: Dict[str, Tensor], x: Tensor, *, training: bool,
                      mc_dropout: bool = False, masks: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """MimoUNet.forward — model.py:94-117.  x [N,S,Ci,H,W] -> [N,S,Co,H,W].

    ``training`` selects batch-statistics BN; dropout is active when
    ``training or mc_dropout`` (ensemble.py:54-66 re-enables Dropout* modules
    in eval mode).  ``masks`` maps DoubleConv prefix -> [N,C] multipliers.
    center/final dropout (elementwise nn.Dropout, model.py:213,277-281) accept
    full-shape multipliers under keys 'core.center_dropout' /
    'decoder.final_dropouts.{s}'.
    """
    masks = masks or {}
    S = cfg.num_subnetworks
    assert x.shape[1] == S and x.shape[2] == cfg.in_channels
    drop = training or mc_dropout

    def dc(prefix, t, p):
        return double_conv(t, st, prefix, training, p, drop, masks.get(prefix))

    x1s, x2s = [], []
    for s in range(S):  # SubnetworkEncoder.forward — model.py:150-175
        x1 = dc(f"encoder.in_convs.{s}.double_conv", x[:, s], cfg.encoder_dropout_rate)
        x2 = dc(f"encoder.down1s.{s}.conv.double_conv", F.max_pool2d(x1, 2), cfg.encoder_dropout_rate)
        x1s.append(x1)
        x2s.append(x2)
    x2c = torch.cat(x2s, dim=1)  # model.py:113

    # SubnetworkCore.forward — model.py:232-243
    p = cfg.core_dropout_rate
    x3 = dc("core.down2.conv.double_conv", F.max_pool2d(x2c, 2), p)
    x4 = dc("core.down3.conv.double_conv", F.max_pool2d(x3, 2), p)
    x5 = dc("core.down4.conv.double_conv", F.max_pool2d(x4, 2), p)
    if drop and cfg.center_dropout_rate > 0:
        m = masks.get("core.center_dropout")
        x5 = F.dropout(x5, cfg.center_dropout_rate, True) if m is None else x5 * m
    u = dc("core.up1.conv.double_conv", up_cat(x5, x4), p)
    u = dc("core.up2.conv.double_conv", up_cat(u, x3), p)
    u = dc("core.up3.conv.double_conv", up_cat(u, x2c), p)

    outs = []
    for s in range(S):  # SubnetworkDecoder.forward — model.py:285-297
        h = dc(f"decoder.up4s.{s}.conv.double_conv", up_cat(u, x1s[s]), cfg.decoder_dropout_rate)
        if drop and cfg.final_dropout_rate > 0:
            m = masks.get(f"decoder.final_dropouts.{s}")
            h = F.dropout(h, cfg.final_dropout_rate, True) if m is None else h * m
        outs.append(F.conv2d(h, st[f"decoder.outcs.{s}.conv.weight"], st[f"decoder.outcs.{s}.conv.bias"]))
    return torch.stack(outs, dim=1)


def split_heads(out: Tensor, out_channels: int) -> Tuple[Tensor, Tensor]:
    """MimoUnetModel.forward split — mimo_unet.py:110-111."""
    h = out_channels // 2
    return out[:, :, :h], out[:, :, h:]


# --------------------------------------------------------------------------
# losses (losses.py:39-192)
# --------------------------------------------------------------------------
EPS_MIN, EPS_MAX = 1e-5, 1e3


def _clamped_no_grad(v: Tensor) -> Tensor:
    v = v.clone()
    with torch.no_grad():
        v.clamp_(min=EPS_MIN, max=EPS_MAX)
    return v


def laplace_nll(y_hat: Tensor, log_scale: Tensor, y: Tensor, mask: Optional[Tensor] = None,
                reduce_mean: bool = True) -> Tensor:
    """LaplaceNLL.forward — losses.py:132-164.  The clamp is applied to the
    value only (in-place under no_grad), so d/dlog_scale keeps exp(log_scale)."""
    diff = y_hat - y
    scale = _clamped_no_grad(torch.exp(log_scale))
    loss = torch.log(scale) + diff.abs() / scale
    if mask is not None:
        loss = loss * mask
    return loss.mean() if reduce_mean else loss


def gaussian_nll(y_hat: Tensor, log_var: Tensor, y: Tensor, mask: Optional[Tensor] = None,
                 reduce_mean: bool = True) -> Tensor:
    """GaussianNLL.forward — losses.py:47-79."""
    diff = y_hat - y
    var = _clamped_no_grad(torch.exp(log_var))
    loss = torch.log(var) + diff ** 2 / var
    if mask is not None:
        loss = loss * mask
    return loss.mean() if reduce_mean else loss


def loss_std(kind: str, log_param: Tensor) -> Tensor:
    """std() — losses.py:82-87 (gaussian), :166-167 (laplace)."""
    if kind == "laplace_nll":
        return torch.exp(log_param) * (2 ** 0.5)
    return torch.exp(log_param) ** 0.5


def calculate_dist_param(kind: str, std: Tensor, log: bool = False) -> Tensor:
    """calculate_dist_param — losses.py:96-121 / :172-192."""
    param = std / (2 ** 0.5) if kind == "laplace_nll" else std ** 2
    param = _clamped_no_grad(param)
    return torch.log(param) if log else param


def loss_forward(kind: str, *a, **k) -> Tensor:
    return laplace_nll(*a, **k) if kind == "laplace_nll" else gaussian_nll(*a, **k)


def laplace_nll_grads(y_hat: Tensor, log_scale: Tensor, y: Tensor) -> Tuple[Tensor, Tensor]:
    """Closed-form per-element gradients of the un-reduced Laplace NLL
    (probe-verified): d/dy_hat = sign(d)/s_c,
    d/dlog_scale = (1/s_c - |d|/s_c^2) * exp(log_scale)."""
    d = y_hat - y
    e = torch.exp(log_scale)
    sc = e.clamp(EPS_MIN, EPS_MAX)
    return torch.sign(d) / sc, (1.0 / sc - d.abs() / (sc * sc)) * e


# --------------------------------------------------------------------------
# MIMO helpers (utils.py:5-101)
# --------------------------------------------------------------------------
def apply_perms(t: Optional[Tensor], perms: Tensor) -> Optional[Tensor]:
    """apply_input_transform with the random draws made explicit: perms [S,N']
    (utils.py:38-48).  t [N,C,H,W] -> [N',S,C,H,W]."""
    if t is None:
        return None
    return torch.stack([t.index_select(0, perms[s]) for s in range(perms.shape[0])], dim=1)


def draw_perms(n: int, num_subnetworks: int, input_repetition_probability: float = 0.0,
               batch_repetitions: int = 1, generator: Optional[torch.Generator] = None) -> Tensor:
    """The index draw of apply_input_transform — utils.py:27-36."""
    main = torch.randperm(n, generator=generator).repeat(batch_repetitions)
    k = int(main.shape[0] * (1.0 - input_repetition_probability))
    return torch.stack([
        torch.cat((main[:k][torch.randperm(k, generator=generator)], main[k:]), dim=0)
        for _ in range(num_subnetworks)
    ])


def repeat_subnetworks(x: Tensor, S: int) -> Tensor:
    return x[:, None].repeat(1, S, 1, 1, 1)  # utils.py:51-61


def compute_uncertainties(kind: str, y_preds: Tensor, log_params: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """compute_uncertainties — utils.py:76-101."""
    S = y_preds.shape[1]
    mean = y_preds.mean(dim=1)
    alea = torch.square(loss_std(kind, log_params)).mean(dim=1)
    if S > 1:
        epi = torch.square(y_preds - y_preds.mean(dim=1, keepdim=True)).sum(dim=1) / (S - 1)
    else:
        epi = torch.zeros_like(alea)
    return mean, alea, epi


class LossBuffer:
    """loss_buffer.py:18-74 — zero-initialised ring; mean over ALL rows."""

    def __init__(self, subnetworks: int, temperature: float, buffer_size: int):
        self.index, self.temperature, self.buffer_size, self.subnetworks = 0, temperature, buffer_size, subnetworks
        self.buffer = torch.zeros(buffer_size, subnetworks)

    def add(self, loss: Tensor) -> None:
        if self.buffer_size != 0:
            self.buffer[self.index] = loss
            self.index = (self.index + 1) % self.buffer_size

    def get_weights(self) -> Tensor:
        mean = self.buffer.mean(dim=0) if self.buffer_size != 0 else torch.zeros(self.subnetworks)
        assert self.temperature > 0
        return F.softmax(mean / self.temperature, dim=-1) * len(mean)


# --------------------------------------------------------------------------
# optimiser (mimo_unet.py:185-201: torch.optim.Adam, L2-in-grad; StepLR)
# --------------------------------------------------------------------------
def adam_update(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
                beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 0.0) -> None:
    """One torch.optim.Adam (not AdamW) update, in place; ``step`` counts from 1."""
    if weight_decay != 0.0:
        g = g + weight_decay * p
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def step_lr(base_lr: float, epoch: int, step_size: int = 20, gamma: float = 0.5) -> float:
    return base_lr * gamma ** (epoch // step_size)


# --------------------------------------------------------------------------
# one full training step (mimo_unet.py:115-144 + Lightning backward + Adam)
# --------------------------------------------------------------------------
@dataclass
class TrainState:
    cfg: NetConfig
    st: Dict[str, Tensor]
    loss_kind: str = "laplace_nll"
    lr: float = 1e-3
    weight_decay: float = 0.0
    loss_buffer: LossBuffer = None  # type: ignore
    step: int = 0
    exp_avg: Dict[str, Tensor] = field(default_factory=dict)
    exp_avg_sq: Dict[str, Tensor] = field(default_factory=dict)

    def param_names(self) -> List[str]:
        return [n for n in self.st if not is_buffer(n)]


def reference_autocast(kind: Optional[str]):
    """The reference's own mixed precision: Lightning `precision="16-mixed"` (scripts/train/train_ndvi.py:71,
    train_nyuv2_depth.py:74) / "bf16-mixed" runs training_step under `torch.autocast(device, dtype)`.  Every operator of
    this restatement is the torch functional the reference's modules call, so running it under the same context
    reproduces the reference's rounding points (pinned by tests/golden/amp_*.npz, generated from the reference itself
    under torch's CPU autocast policy: conv2d in the 16-bit type, reflection_pad2d in fp32, BatchNorm on 16-bit
    activations with fp32 statistics, 16-bit logits, loss by type promotion).  kind: "bf16-mixed" | "16-mixed" | None."""
    if kind is None:
        return contextlib.nullcontext()
    return torch.autocast("cpu", dtype={"bf16-mixed": torch.bfloat16, "16-mixed": torch.float16}[kind])


def train_step(ts: TrainState, image: Tensor, label: Tensor, mask: Optional[Tensor], perms: Tensor,
               masks: Optional[Dict[str, Tensor]] = None, apply_optimizer: bool = True,
               want_input_grad: bool = False, loss_scale: float = 1.0, autocast: Optional[str] = None) -> Dict[str, Tensor]:
    """training_step + backward + Adam.  Returns losses, weights, grads, outputs.  loss_scale: the backward runs on
    loss_scale * total and the gradients are divided by it (what a GradScaler does; matters only with the 16-bit
    gradient-storage emulation).  autocast: run forward + loss under `reference_autocast(kind)`, the backward outside it
    (Lightning's MixedPrecisionPlugin)."""
    cfg = ts.cfg
    names = ts.param_names()
    leaves = {n: ts.st[n].detach().clone().requires_grad_(True) for n in names}
    st = dict(ts.st)
    st.update(leaves)
    x = apply_perms(image, perms)
    if want_input_grad:
        x = x.detach().requires_grad_(True)
    y = apply_perms(label, perms)
    mk = apply_perms(mask, perms)
    with reference_autocast(autocast):
        out = mimo_unet_forward(cfg, st, x, training=True, masks=masks)
        for n in ts.st:  # BN running buffers were updated through the st copy
            if is_buffer(n):
                ts.st[n] = st[n]
        p1, p2 = split_heads(out, cfg.out_channels)
        per_elem = loss_forward(ts.loss_kind, p1, p2, y, mask=mk, reduce_mean=False)
        loss = per_elem.mean(dim=(0, 2, 3, 4))                     # mimo_unet.py:241-242
        weights = ts.loss_buffer.get_weights()                      # read BEFORE add (:243-245)
        ts.loss_buffer.add(loss.detach())
        total = (loss * weights).mean()                             # :138
    (total * loss_scale).backward()
    grads = {n: leaves[n].grad / loss_scale for n in names}
    res = {"out": out.detach(), "loss": loss.detach(), "weights": weights, "total": total.detach(), "grads": grads}
    if want_input_grad:
        res["dx"] = x.grad / loss_scale
    if apply_optimizer:
        ts.step += 1
        for n in names:
            if n not in ts.exp_avg:
                ts.exp_avg[n] = torch.zeros_like(ts.st[n])
                ts.exp_avg_sq[n] = torch.zeros_like(ts.st[n])
            adam_update(ts.st[n], grads[n], ts.exp_avg[n], ts.exp_avg_sq[n], ts.step, ts.lr,
                        weight_decay=ts.weight_decay)
    return res


def ensemble_forward(cfg: NetConfig, st: Dict[str, Tensor], x: Tensor, loss_kind: str = "laplace_nll",
                     monte_carlo_steps: int = 0, pass_masks: Optional[Sequence[Dict[str, Tensor]]] = None,
                     raw: bool = False):
    """EnsembleModule.forward for ONE checkpoint — ensemble.py:76-115: eval-mode
    BN, dropout re-enabled when monte_carlo_steps > 0, passes concatenated on
    the subnetwork axis, then compute_uncertainties."""
    xr = repeat_subnetworks(x, cfg.num_subnetworks)
    p1s, p2s = [], []
    for i in range(max(1, monte_carlo_steps)):
        with torch.no_grad():
            out = mimo_unet_forward(cfg, st, xr, training=False, mc_dropout=monte_carlo_steps > 0,
                                    masks=None if pass_masks is None else pass_masks[i])
        a, b = split_heads(out, cfg.out_channels)
        p1s.append(a)
        p2s.append(b)
    p1, p2 = torch.cat(p1s, dim=1), torch.cat(p2s, dim=1)
    return (p1, p2) if raw else compute_uncertainties(loss_kind, p1, p2)


# ---------------------------------------------------------------------------------------
# Evidential regression variant (mimo/models/evidential_unet.py:74-96, mimo/losses.py:195-271)
# ---------------------------------------------------------------------------------------
def evidential_forward(cfg: NetConfig, st: Dict[str, Tensor], x: Tensor, *, training: bool,
                       masks: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """EvidentialUnetModel.forward — evidential_unet.py:74-96.  x [B,Ci,H,W] -> [B,4,H,W] =
    (gamma, v, alpha, beta); the backbone is the S=1 MIMO U-Net with four output channels."""
    assert cfg.num_subnetworks == 1 and cfg.out_channels == 4
    out = mimo_unet_forward(cfg, st, x[:, None], training=training, masks=masks)[:, 0]
    mu, logv, logalpha, logbeta = torch.unbind(out, dim=1)
    return torch.stack([mu, F.softplus(logv), F.softplus(logalpha) + 1, F.softplus(logbeta)], dim=1)


def evidential_loss(ev: Tensor, y: Tensor, mask: Optional[Tensor] = None) -> Tensor:
    """EvidentialLoss.forward(reduce_mean=False) — losses.py:202-247: sum-of-squares NIG loss plus the
    evidence regulariser, per pixel [B,H,W]."""
    mu, v, alpha, beta = torch.unbind(ev, dim=1)
    t = y.squeeze(1)
    coeff = torch.exp(torch.lgamma(alpha - 0.5)) / (4 * torch.exp(torch.lgamma(alpha)) * v * torch.sqrt(beta))
    sos = coeff * (2 * beta * (1 + v) + (2 * alpha - 1) * v * (t - mu) ** 2)
    reg = (t - mu) ** 2 * (2 * alpha + v)
    loss = sos + reg
    return loss * mask if mask is not None else loss


def evidential_vars(ev: Tensor) -> Tuple[Tensor, Tensor]:
    """aleatoric_var, epistemic_var — losses.py:259-271."""
    _, v, alpha, beta = torch.unbind(ev, dim=1)
    return beta / (alpha - 1), beta / (v * (alpha - 1))
