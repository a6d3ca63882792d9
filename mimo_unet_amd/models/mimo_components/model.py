"""`MimoUNet` with the reference's constructor, tensor contract and state_dict, executed by
libmimo_hip.so.

Interface mirrored: ``mimo/models/mimo_components/model.py:26-117`` (``MimoUNet``) of the
reference — ``forward([B,S,C_in,H,W]) -> [B,S,C_out,H,W]``, ``parameters()``,
``state_dict()`` with the reference's names and OIHW shapes, ``modules()`` exposing the
``Dropout*`` children that ``EnsembleModule._activate_mc_dropout`` toggles
(``mimo/models/ensemble.py:54-66``).

The nn.Module tree below only *holds* parameters (stock torch layers used as named
containers, constructed in the reference's order so a given seed yields the same
initial values); no torch layer is ever called.  Execution: parameters live in one flat
device buffer bound to a `mimo_plan`; forward / backward are single C-ABI calls wrapped
in a `torch.autograd.Function`."""
from __future__ import annotations

import logging
import os
import weakref
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import torch
from torch import nn

from ... import _lib as L
from ...engine import NetGeometry, Plan

logger = logging.getLogger(__name__)


# ----------------------------------------------------------------------------------------
# parameter holders (names == reference state_dict names)
# ----------------------------------------------------------------------------------------
def _conv_bn_relu(cin: int, cout: int) -> List[nn.Module]:
    return [nn.Conv2d(cin, cout, kernel_size=3, padding=1, padding_mode="reflect"), nn.BatchNorm2d(cout),
            nn.ReLU(inplace=True)]


class DoubleConv(nn.Module):
    """Holder for (conv3x3 reflect -> BN -> ReLU) x 2 -> Dropout2d."""

    def __init__(self, in_channels: int, out_channels: int, dropout_rate: float = 0.0, mid_channels: Optional[int] = None):
        super().__init__()
        mid = mid_channels or out_channels
        self.double_conv = nn.Sequential(*_conv_bn_relu(in_channels, mid), *_conv_bn_relu(mid, out_channels),
                                         nn.Dropout2d(dropout_rate))

    @property
    def dropout(self) -> nn.Dropout2d:
        return self.double_conv[6]

    @property
    def norm(self) -> nn.BatchNorm2d:
        return self.double_conv[1]


class Down(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, dropout_rate: float = 0.0):
        super().__init__()
        self.conv = DoubleConv(in_channels, out_channels, dropout_rate=dropout_rate)


class Up(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, dropout_rate: float = 0.0):
        super().__init__()
        self.conv = DoubleConv(in_channels, out_channels, dropout_rate=dropout_rate, mid_channels=in_channels // 2)


class OutConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)


class SubnetworkEncoder(nn.Module):
    def __init__(self, S: int, in_channels: int, f: int, p: float):
        super().__init__()
        self.in_convs = nn.ModuleList([DoubleConv(in_channels, f, dropout_rate=p) for _ in range(S)])
        self.down1s = nn.ModuleList([Down(f, 2 * f, dropout_rate=p) for _ in range(S)])


class SubnetworkCore(nn.Module):
    def __init__(self, S: int, f: int, p: float, center_p: float):
        super().__init__()
        w = f * S
        self.down2 = Down(2 * w, 4 * w, dropout_rate=p)
        self.down3 = Down(4 * w, 8 * w, dropout_rate=p)
        self.down4 = Down(8 * w, 8 * w, dropout_rate=p)
        self.center_dropout = nn.Dropout(p=center_p)
        self.up1 = Up(16 * w, 4 * w, dropout_rate=p)
        self.up2 = Up(8 * w, 2 * w, dropout_rate=p)
        self.up3 = Up(4 * w, w, dropout_rate=p)


class SubnetworkDecoder(nn.Module):
    def __init__(self, S: int, f: int, out_channels: int, p: float, final_p: float):
        super().__init__()
        self.up4s = nn.ModuleList([Up(f * S + f, f, dropout_rate=p) for _ in range(S)])
        self.final_dropouts = nn.ModuleList([nn.Dropout(p=final_p) for _ in range(S)])
        self.outcs = nn.ModuleList([OutConv(f, out_channels) for _ in range(S)])


# ----------------------------------------------------------------------------------------
# autograd bridge
# ----------------------------------------------------------------------------------------
class _NetFunction(torch.autograd.Function):
    """out (and optionally the per-subnetwork loss vector) = net(x); gradients through the C ABI."""

    @staticmethod
    def forward(ctx, net: "MimoUNet", x, label, lmask, perm, masks, bn_training, elem_masks, rng, *params):
        ctx.set_materialize_grads(False)
        plan = net._plan_for(x, perm, for_autograd=True)
        n = plan.batch
        S, Co = net.num_subnetworks, net.out_channels
        out = torch.empty(n, S, Co, plan.height, plan.width, device=x.device, dtype=torch.float32)
        plan.bind(net._flat_params, net._flat_grads, net._flat_buffers)
        plan.forward(x, out, training=bn_training, perm=perm, masks=masks, elem_masks=elem_masks, rng=rng,
                     param_version=net._param_version())
        if label is not None:
            loss = torch.empty(S, device=x.device, dtype=torch.float32)
            plan.loss_forward(label, lmask, perm, loss)
        else:
            loss = torch.zeros(0, device=x.device, dtype=torch.float32)
        plan.generation += 1
        ctx.net, ctx.plan, ctx.generation = net, plan, plan.generation
        # device memory the plan still points at; the output goes through save_for_backward (keeping it in a plain
        # attribute would tie ctx -> out -> grad_fn -> ctx into a cycle that only the garbage collector breaks)
        ctx.keep = (x, label, lmask, perm, masks, elem_masks)
        ctx.save_for_backward(out)
        # the plan holds the saved activations of THIS graph until its backward has run or the graph is dropped; a
        # forward of the same geometry in between gets another plan (_plan_for), like autograd keeping two graphs alive
        plan.pending = plan.generation
        weakref.finalize(ctx, _release_plan, plan, plan.generation)
        ctx.x_shape = x.shape
        ctx.has_loss = label is not None
        ctx.mark_non_differentiable(*[])
        return out, loss

    @staticmethod
    def backward(ctx, dout, dloss):
        net, plan = ctx.net, ctx.plan
        n_in = 9 + len(net._param_list)
        if dout is None and (dloss is None or not ctx.has_loss):
            return (None,) * n_in
        if plan.generation != ctx.generation:
            raise RuntimeError("MimoUNet: backward() after a newer forward() of the same shape — the engine keeps one "
                               "set of saved activations per input geometry")
        x = ctx.keep[0]
        dx = None
        if ctx.needs_input_grad[1]:
            if ctx.keep[3] is not None or x.dim() != 5:
                raise RuntimeError("MimoUNet: input gradients need the [B,S,C,H,W] forward without a fused permutation")
            dx = torch.empty(ctx.x_shape, device=x.device, dtype=torch.float32)
        dout_c = None if dout is None else dout.contiguous().float()
        dloss_c = None if (dloss is None or not ctx.has_loss) else dloss.contiguous().float()
        net._run_backward(plan, dout_c, dloss_c, dx)
        _release_plan(plan, ctx.generation)
        return (None, dx, None, None, None, None, None, None, None) + (None,) * len(net._param_list)


def _release_plan(plan, generation: int) -> None:
    if getattr(plan, "pending", None) == generation:
        plan.pending = None


def _guard_after_load(module, incompatible_keys) -> None:
    module.guard_fp16_range()


class MimoUNet(nn.Module):
    """Multiple-input multiple-output U-Net: S private encoders -> shared core on the channel
    concat -> S private decoders/heads.  Same constructor as the reference's class."""

    def __init__(self, in_channels: int, out_channels: int, num_subnetworks: int, filter_base_count: int = 30,
                 center_dropout_rate: float = 0.0, final_dropout_rate: float = 0.0, encoder_dropout_rate: float = 0.0,
                 core_dropout_rate: float = 0.0, decoder_dropout_rate: float = 0.0, bilinear: bool = True,
                 use_pooling_indices: bool = False, loss: str = "laplace_nll"):
        spatial = encoder_dropout_rate > 0.0 or core_dropout_rate > 0.0 or decoder_dropout_rate > 0.0
        if spatial and (center_dropout_rate > 0.0 or final_dropout_rate > 0.0):
            raise ValueError("Do not specify spatial_dropout together with center_dropout_rate or final_dropout_rate!")
        if not bilinear or use_pooling_indices:
            raise NotImplementedError("only bilinear=True, use_pooling_indices=False (what MimoUnetModel hard-wires)")
        super().__init__()
        logger.info("Creating MimoUNet(HIP): in=%d out=%d S=%d f=%d dropout enc/core/dec=%g/%g/%g", in_channels, out_channels,
                    num_subnetworks, filter_base_count, encoder_dropout_rate, core_dropout_rate, decoder_dropout_rate)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_subnetworks, self.filter_base_count = num_subnetworks, filter_base_count
        S, f = num_subnetworks, filter_base_count
        self.encoder = SubnetworkEncoder(S, in_channels, f, encoder_dropout_rate)
        self.core = SubnetworkCore(S, f, core_dropout_rate, center_dropout_rate)
        self.decoder = SubnetworkDecoder(S, f, out_channels, decoder_dropout_rate, final_dropout_rate)
        # arithmetic of the 3x3 forward / data-gradient convolutions (include/mimo_hip.h mimo_precision)
        precision = os.environ.get("MIMO_PRECISION", "split16")
        self._geom = NetGeometry(in_channels, out_channels, S, f, encoder_dropout_rate, core_dropout_rate,
                                 decoder_dropout_rate, center_dropout_rate, final_dropout_rate, loss, precision)
        # Dropout masks are drawn inside the engine (Philox stream keyed by torch's CUDA generator: torch.manual_seed
        # governs them); False: drawn with torch.bernoulli on the device and handed over as tensors
        self.engine_rng = os.environ.get("MIMO_ENGINE_RNG", "1") != "0"
        # data-parallel backward: stages without a per-stage join of the engine's two streams (engine.Plan.backward)
        self.async_stages = os.environ.get("MIMO_DDP_ASYNC_STAGES", "1") != "0"
        # one plan per (batch, H, W, device, inference-only); least recently used plans are dropped beyond
        # MIMO_PLAN_CACHE entries (a plan owns its whole activation workspace: ragged last batches, separate
        # train / val batch sizes and variable image sizes would otherwise pile up multi-GB plans)
        self._plans: "OrderedDict[tuple, Plan]" = OrderedDict()
        self._plan_cache_size = max(1, int(os.environ.get("MIMO_PLAN_CACHE", "4")))
        self._flat_params = self._flat_grads = self._flat_buffers = None
        self._param_list: List[nn.Parameter] = []
        self._versioned: List[torch.Tensor] = []
        self._flat_device = None
        self.mask_override: Optional[Dict[int, torch.Tensor]] = None  # tests: {double-conv index: [N,C] multipliers}
        # tests: {"center" | "final{s}": full-shape nn.Dropout multipliers (reference NCHW layout)}
        self.elem_mask_override: Optional[Dict[str, torch.Tensor]] = None
        # called as hook(flat_grads, begin, end) when gradients [begin, end) are final (see ddp.FlatGradientAllReducer)
        self.grad_ready_hook = None
        # bumped by everything that rewrites the flat parameter / buffer storage behind torch's back
        # (FlatAdam's HIP kernel, the engine's running-stat update); together with the tensors' own
        # torch version counters it forms mimo_forward_args.param_version
        self._param_epoch = 1
        self.precision_guard = None  # (from, to, bound) once guard_fp16_range() has switched the precision mode
        # after a (partial) load of this subtree: do the loaded BatchNorm parameters fit the fp16 operands of the mode?
        self.register_load_state_dict_post_hook(_guard_after_load)

    # ---- copies / pickles ----------------------------------------------------------------------
    # Everything the engine derives from the parameters is transient: plans (ctypes handles into libmimo_hip.so), the flat
    # storage the parameters are views of, cached module lists, hooks of a data-parallel reducer.  `copy.deepcopy(model)`
    # (EMA / SWA callbacks) and `torch.save(model)` take the module tree with its parameters and buffers; the copy rebuilds
    # the rest at its first forward.
    _TRANSIENT = {"_plans": None, "_flat_params": None, "_flat_grads": None, "_flat_buffers": None, "_flat_counters": None,
                  "_flat_device": None, "_flat_probe": None, "_param_list": None, "_param_views": None, "_versioned": None,
                  "_dc_cache": None, "_drop_cache": None, "_mask_plan": None, "_inference_keep": None, "grad_ready_hook": None,
                  "grad_sync": None}

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in self._TRANSIENT:
            state.pop(k, None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__.update({"_plans": OrderedDict(), "_flat_params": None, "_flat_grads": None, "_flat_buffers": None,
                              "_flat_device": None, "_param_list": [], "_versioned": [], "grad_ready_hook": None})

    # ---- execution order of the DoubleConvs == mimo_plan's (engine) order -----------------
    def double_convs(self) -> List[DoubleConv]:
        dcs = self.__dict__.get("_dc_cache")
        if dcs is None:  # (a plain attribute, not a registered submodule list: the modules stay owned by the tree above)
            c = self.core
            dcs = ([m for m in self.encoder.in_convs] + [d.conv for d in self.encoder.down1s]
                   + [c.down2.conv, c.down3.conv, c.down4.conv, c.up1.conv, c.up2.conv, c.up3.conv]
                   + [u.conv for u in self.decoder.up4s])
            self.__dict__["_dc_cache"] = dcs
            self.__dict__["_drop_cache"] = ([dc.double_conv[6] for dc in dcs] + [self.core.center_dropout]
                                            + list(self.decoder.final_dropouts))
        return dcs

    def _dropout_modules(self) -> List[nn.Module]:
        """Dropout2d of every DoubleConv in engine order, then center_dropout, then final_dropouts[s]"""
        self.double_convs()
        return self.__dict__["_drop_cache"]

    # ---- fp16 range guard (round 6) --------------------------------------------------------------------------------------
    # "split16" / "16-mixed" carry activations as fp16 operands: |a| >= 65 520 overflows where the reference's fp32
    # convolution (components.py:23,26) stays finite.  Every activation that reaches a convolution is a BatchNorm + ReLU output
    # a = relu(gamma * xhat + beta) (times a dropout multiplier), so the PARAMETERS bound it: |a| <= |gamma| * |xhat| + |beta|.
    # The guard takes |xhat| <= kGuardXhat = 256 standard deviations — beyond anything a normalised tensor of real data
    # holds, far below the sqrt(P) a single outlier can reach in theory — and when max(|gamma| * 256 + |beta|) / (1 - p) reaches
    # fp16's range (|gamma| >= ~250: a diverged or hand-made checkpoint) it moves the WHOLE network to the mode with fp32 exponent
    # range ("fp32" from "split16", "bf16-mixed" from "16-mixed"), with a warning.  Coarse on purpose: the per-layer remedy
    # (a power-of-two activation scale per layer, like the weights') is not built.  Evaluated when a
    # state_dict has been loaded and in check_numerics() (the Lightning epoch-end hooks); MIMO_FP16_RANGE_GUARD=0: off.
    kGuardXhat = 256.0

    def fp16_activation_bound(self) -> float:
        bns = [dc.double_conv[i] for dc in self.double_convs() for i in (1, 4)]
        with torch.no_grad():
            tops = [(bn.weight.detach().abs() * self.kGuardXhat + bn.bias.detach().abs()).max() for bn in bns]
            bound = float(torch.stack([t.float().cpu() if not t.is_cuda else t.float() for t in tops]).max())
        p = max([d.p for d in self._dropout_modules()] + [0.0])
        return bound / max(1.0 - p, 1e-3)

    def guard_fp16_range(self) -> bool:
        """True when the guard moved the network to a mode with fp32 exponent range (see above)."""
        if os.environ.get("MIMO_FP16_RANGE_GUARD", "1") == "0" or self._geom.precision not in ("split16", "16-mixed"):
            return False
        bound = self.fp16_activation_bound()
        if not bound >= 65504.0:  # (a NaN bound is a diverged run: check_numerics reports that)
            return False
        to = "fp32" if self._geom.precision == "split16" else "bf16-mixed"
        logger.warning("MimoUNet: BatchNorm parameters allow activations up to %.3g (|gamma| * %g + |beta|), beyond the fp16 "
                       "operands of precision %r; switching to %r (fp32 exponent range). MIMO_FP16_RANGE_GUARD=0 disables this.",
                       bound, self.kGuardXhat, self._geom.precision, to)
        self.precision_guard = (self._geom.precision, to, bound)
        self.set_precision(to)
        return True

    def set_precision(self, precision: str) -> None:
        """"fp32" (f32-input MFMA, exact) or "split16" (split-bf16 MFMA, ~1e-5 relative per product)."""
        if precision != self._geom.precision:
            self._geom = NetGeometry(**{**self._geom.__dict__, "precision": precision})
            for plan in self._plans.values():  # what the dropped plans recorded outlives them (numerics_status ORs it in)
                self._evicted_status = getattr(self, "_evicted_status", 0) | plan.status(True)
            self._plans.clear()

    def set_loss(self, loss: str) -> None:
        if loss != self._geom.loss:
            self._geom = NetGeometry(**{**self._geom.__dict__, "loss": loss})
            self._plans.clear()

    # ---- flat storage ---------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._flat_params = None  # .to()/.cuda() re-created the tensors: re-flatten lazily
        self._plans = OrderedDict()
        return r

    def _ensure_flat(self, plan: Plan, device) -> None:
        # (every step passes through here twice: the test must not walk the module tree — dict(named_parameters()) +
        # dict(named_buffers()) per call were 1.2 of the 4.1 ms the host needed per step at 4 images per GPU, round 6)
        if self._flat_params is not None and self._flat_device == device:
            owner, attr, off = self._flat_probe
            if getattr(owner, attr).data_ptr() == self._flat_params.data_ptr() + 4 * off:
                return
        named_p = dict(self.named_parameters())
        named_b = dict(self.named_buffers())
        first = plan.specs[0]
        fp = torch.zeros(plan.param_floats, device=device, dtype=torch.float32)
        fb = torch.zeros(plan.buffer_floats, device=device, dtype=torch.float32)
        plist = []
        for sp in plan.specs:
            src = named_p[sp.name] if sp.kind == 0 else named_b[sp.name]
            if tuple(src.shape) != sp.shape:
                raise RuntimeError(f"{sp.name}: shape {tuple(src.shape)} != engine shape {sp.shape}")
            flat = fp if sp.kind == 0 else fb
            view = flat[sp.offset: sp.offset + sp.numel].view(sp.shape)
            view.copy_(src.data)
            src.data = view
            if sp.kind == 0:
                plist.append(src)
        missing = set(named_p) - {sp.name for sp in plan.specs}
        if missing:
            raise RuntimeError(f"parameters unknown to the engine: {sorted(missing)[:4]}")
        self._flat_params, self._flat_buffers = fp, fb
        self._flat_grads = torch.zeros_like(fp)
        self._param_list = plist
        self._param_views = [(p, self._flat_grads[sp.offset: sp.offset + sp.numel].view(sp.shape))
                             for p, sp in zip(plist, [s for s in plan.specs if s.kind == 0])]
        self._flat_device = device
        self._versioned = plist + [named_b[sp.name] for sp in plan.specs if sp.kind == 1]
        # the probe of the early-out above: the module that owns the first engine tensor, looked up by attribute each time
        # (a parameter object that was REPLACED — `conv.weight = nn.Parameter(...)` — is seen, like the dict lookup saw it)
        mod_name, _, attr = first.name.rpartition(".")
        self._flat_probe = (self.get_submodule(mod_name) if mod_name else self, attr, first.offset)
        # BatchNorm's num_batches_tracked counters as views of ONE int64 tensor: a training forward bumps them with one
        # add_ (torch._foreach_add_ over the 24 scalar tensors cost the host 0.67 ms per step)
        bns = [bn for dc in self.double_convs() for bn in (dc.double_conv[1], dc.double_conv[4])]
        self._flat_counters = torch.stack([bn.num_batches_tracked.to(device=device, dtype=torch.int64) for bn in bns])
        for i, bn in enumerate(bns):
            bn.num_batches_tracked.data = self._flat_counters[i]

    def _plan_for(self, x: torch.Tensor, perm: Optional[torch.Tensor], inference: bool = False,
                  for_autograd: bool = False) -> Plan:
        """inference: a plan without the buffers only a backward needs (pre-activation tensors, activation
        gradients, dz / padded-gradient / weight-gradient scratch, data-gradient weight copies) — what eval
        mode under torch.no_grad() uses, e.g. the passes x B samples of an MC-dropout ensemble."""
        if not x.is_cuda:
            raise L.MimoHipError("MimoUNet runs on an AMD GPU through libmimo_hip.so; move the module and its inputs "
                                 "to cuda (there is no CPU execution path)")
        n = perm.shape[1] if perm is not None else x.shape[0]
        key = (n, x.shape[-2], x.shape[-1], x.device.index, bool(inference))
        plan = self._plans.get(key)
        if for_autograd and plan is not None and getattr(plan, "pending", None) is not None:
            # an earlier graph of this geometry is still alive and may yet be back-propagated (two forwards, then the
            # first one's backward — torch's autograd allows it): use further plans of the same geometry (key + slot)
            slot = 1
            while True:
                k2 = key + (slot,)
                p2 = self._plans.get(k2)
                if p2 is None or getattr(p2, "pending", None) is None:
                    key, plan = k2, p2
                    break
                slot += 1
        if plan is None:
            while len(self._plans) >= self._plan_cache_size:
                # a pending autograd node keeps its own reference to its plan.  What the evicted plan recorded in its
                # numerics status word (a NaN on a ragged last batch, say) outlives it: numerics_status() ORs it in
                _, old = self._plans.popitem(last=False)
                self._evicted_status = getattr(self, "_evicted_status", 0) | old.status(True)
            plan = Plan(self._geom, n, x.shape[-2], x.shape[-1], x.device, inference_only=inference)
            plan.generation = 0
            plan.pending = None
            self._plans[key] = plan
        else:
            self._plans.move_to_end(key)
        self._ensure_flat(plan, x.device)
        return plan

    # ---- dropout masks (Dropout2d: one Bernoulli per (sample, channel), components.py:29) --
    def _dropout_masks(self, n: int, device) -> Optional[List[Optional[torch.Tensor]]]:
        dcs = self.double_convs()
        if self.mask_override is not None:
            masks, any_mask = [], False
            for i, dc in enumerate(dcs):
                d = dc.dropout
                m = None
                if i in self.mask_override:
                    m = self.mask_override[i].to(device=device, dtype=torch.float32).contiguous()
                elif d.p > 0.0 and d.training:
                    c = dc.double_conv[3].out_channels
                    m = torch.bernoulli(torch.full((n, c), 1.0 - d.p, device=device)).div_(1.0 - d.p)
                any_mask |= m is not None
                masks.append(m)
            return masks if any_mask else None
        # all sites in two launches: one Bernoulli draw over a flat [site][n][C] buffer with per-element keep
        # probabilities, one multiply by 1/(1-p); the per-site masks are contiguous views of it
        active = tuple((d.p if d.training else 0.0) for d in self._dropout_modules()[:len(dcs)])
        if not any(p > 0.0 for p in active):
            return None
        key = (n, str(device), active)
        cache = getattr(self, "_mask_plan", None)
        if cache is None or cache[0] != key:
            chans = [dc.double_conv[3].out_channels for dc in dcs]
            offs, total = [], 0
            for c, p in zip(chans, active):
                offs.append(total if p > 0.0 else -1)
                total += n * c if p > 0.0 else 0
            keep = torch.empty(total, device=device, dtype=torch.float32)
            for c, p, o in zip(chans, active, offs):
                if o >= 0:
                    keep[o: o + n * c] = 1.0 - p
            cache = (key, keep, 1.0 / keep, offs, chans)
            self._mask_plan = cache
        _, keep, inv_keep, offs, chans = cache
        flat = torch.bernoulli(keep).mul_(inv_keep)
        return [flat[o: o + n * c].view(n, c) if o >= 0 else None for o, c in zip(offs, chans)]

    # ---- element-wise dropout (nn.Dropout: center after down4, final in front of each head) ----
    def _elem_dropout_masks(self, n: int, h: int, w: int, device) -> Optional[List[Optional[torch.Tensor]]]:
        S, f = self.num_subnetworks, self.filter_base_count
        elem = self._dropout_modules()[len(self.double_convs()):]
        if self.elem_mask_override is None and not any(d.p > 0.0 and d.training for d in elem):
            return None
        sites = [("center", elem[0], (n, 8 * f * S, h // 16, w // 16))]
        sites += [(f"final{s}", d, (n, f, h, w)) for s, d in enumerate(elem[1:])]
        out, any_mask = [], False
        for key, d, shape in sites:
            m = None
            if self.elem_mask_override is not None and key in self.elem_mask_override:
                m = self.elem_mask_override[key].to(device=device, dtype=torch.float32).contiguous()
                if tuple(m.shape) != shape:
                    raise ValueError(f"{key} dropout mask: expected {shape}, got {tuple(m.shape)}")
            elif d.p > 0.0 and d.training:
                m = torch.bernoulli(torch.full(shape, 1.0 - d.p, device=device)).div_(1.0 - d.p)
            any_mask |= m is not None
            out.append(m)
        return out if any_mask else None

    def _engine_rng_sites(self, device):
        """(sites, seed, offset) for the in-engine generator, or None when no dropout module is active / recorded
        masks are injected.  Advances torch's CUDA generator offset like a torch dropout call would."""
        if not self.engine_rng or self.mask_override is not None or self.elem_mask_override is not None:
            return None
        sites = [bool(d.p > 0.0 and d.training) for d in self._dropout_modules()]
        if not any(sites):
            return None
        gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
        seed, offset = gen.initial_seed(), gen.get_offset()
        gen.set_offset(offset + 4)  # one Philox counter block per forward: every site / element has its own sub-stream
        return sites, seed, offset

    def _bn_training(self) -> bool:
        return self.double_convs()[0].double_conv[1].training

    def _bump_batch_counters(self) -> None:
        self._flat_counters.add_(1)  # the 2 x #DoubleConv counters are views of this tensor (_ensure_flat)

    # ---- forward ----------------------------------------------------------------------------
    def _call(self, x, label, lmask, perm):
        x = x.contiguous().float()
        n = perm.shape[1] if perm is not None else x.shape[0]
        bn_training = self._bn_training()
        rng = self._engine_rng_sites(x.device) if x.is_cuda else None
        masks = None if rng is not None else self._dropout_masks(n, x.device)
        elem_masks = None if rng is not None else self._elem_dropout_masks(n, x.shape[-2], x.shape[-1], x.device)
        # make sure the flat storage exists before the parameters are handed to autograd
        inference = not bn_training and not torch.is_grad_enabled()
        plan = self._plan_for(x, perm, inference=inference)
        if inference:
            # inference (eval mode under torch.no_grad()): no autograd node, BatchNorm + ReLU folded into the
            # convolution epilogue, packed weights reused while the parameters have not changed
            out = torch.empty(n, self.num_subnetworks, self.out_channels, plan.height, plan.width, device=x.device,
                              dtype=torch.float32)
            plan.bind(self._flat_params, self._flat_grads, self._flat_buffers)
            plan.forward(x, out, training=False, perm=perm, masks=masks, elem_masks=elem_masks, no_grad=True,
                         param_version=self._param_version(), rng=rng)
            plan.generation += 1
            self._inference_keep = (x, perm, masks, elem_masks)  # the plan still points at the mask tensors
            if label is None:
                return out, torch.zeros(0, device=x.device, dtype=torch.float32)
            loss = torch.empty(self.num_subnetworks, device=x.device, dtype=torch.float32)
            plan.loss_forward(label, lmask, perm, loss)  # per-subnetwork mean NLL (validation)
            return out, loss
        out, loss = _NetFunction.apply(self, x, label, lmask, perm, masks, bn_training, elem_masks, rng, *self._param_list)
        if bn_training:
            self._bump_batch_counters()
            self._param_epoch += 1  # the engine updated the running statistics in place
        return out, loss

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [B,S,C_in,H,W] -> predictions [B,S,C_out,H,W]."""
        if x.dim() != 5 or x.shape[1] != self.num_subnetworks or x.shape[2] != self.in_channels:
            raise ValueError(f"expected [B,{self.num_subnetworks},{self.in_channels},H,W], got {tuple(x.shape)}")
        return self._call(x, None, None, None)[0]

    def forward_with_loss(self, image: torch.Tensor, label: torch.Tensor, mask: Optional[torch.Tensor],
                          perms: Optional[torch.Tensor]):
        """Fused training entry: `image` [B,C,H,W] gathered per subnetwork through `perms` [S,B'] inside
        the first kernel (or [B,S,C,H,W] with perms=None), plus the per-subnetwork mean NLL [S]."""
        label = label.contiguous().float()
        mask = None if mask is None else mask.contiguous().float()
        perms = None if perms is None else perms.to(device=image.device, dtype=torch.int64).contiguous()
        return self._call(image, label, mask, perms)

    def _run_backward(self, plan: Plan, dout, dloss, dx) -> None:
        g = self._flat_grads
        views = self._param_views
        # torch semantics: gradients accumulate until zero_grad().  The engine overwrites the flat
        # buffer, so keep a copy when a live .grad still aliases it.
        aliased = any(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in views)
        hook = self.grad_ready_hook
        sync = getattr(self, "grad_sync", None)  # the reducer behind the hook (FlatGradientAllReducer.attach), if any
        announcing = hook is not None and (sync is None or sync.enabled)
        # "Already reduced" = the reducer really took ranges of this buffer (not a one-rank no-op reducer) and nothing has
        # written the .grad tensors through torch since: zero_grad(set_to_none=False) of a stock optimiser zeroes them in
        # place, which bumps their version counters — that is a fresh window, not an accumulation.
        # (every view of the flat buffer shares its version counter; the reducer notes it after each collective it issues)
        if aliased and announcing and sync is not None and sync.reduced_version(g) == g._version:
            # data-parallel gradient accumulation done wrong: the gradients of an earlier micro-batch were already
            # summed over the ranks in place; adding them to this micro-batch and reducing again would count them
            # world_size times
            raise RuntimeError("MimoUNet: a second backward accumulates into gradients that were already all-reduced; run "
                               "every micro-batch but the last under FlatGradientAllReducer.no_sync() (mimo_unet_amd.ddp)")
        if aliased and sync is not None and sync.busy:
            sync.finish()  # nothing may still be reducing the buffer the clone below reads and the backward rewrites
        saved = g.clone() if aliased else None
        plan.bind(self._flat_params, g, self._flat_buffers)
        if hook is None:
            plan.backward(dout, dloss, dx)
        elif aliased or not announcing:
            # gradient accumulation (a live .grad aliases the flat buffer): the fix-up below adds the saved
            # gradients into the same memory an in-flight all-reduce would be reducing, so the ranges are only
            # announced once they are final — no overlap on accumulating micro-batches
            plan.backward(dout, dloss, dx)
        else:
            # data-parallel overlap: one stage per core block (heads + decoders first, the encoders last); each
            # stage's slice of the flat gradient buffer is final when it returns and its all-reduce runs while
            # the later stages back-propagate
            # (MIMO_DDP_ASYNC_STAGES, default on: a stage does not make this stream wait for the side stream's weight
            # gradients; its range is final on the stream the engine names, and the collective is issued THERE)
            for st, (b, e) in enumerate(plan.backward_stages):
                ready = plan.backward(dout, dloss, dx, stage=st, async_stage=self.async_stages)
                if ready is None:
                    hook(g, b, e)
                else:
                    with torch.cuda.stream(ready):
                        hook(g, b, e)
        for p, v in views:
            if p.grad is None:
                p.grad = v
            elif p.grad.data_ptr() == v.data_ptr():
                off = (v.data_ptr() - g.data_ptr()) // 4
                v.add_(saved[off: off + v.numel()].view(v.shape))
            else:
                p.grad.add_(v)
        if announcing and aliased:
            for b, e in plan.backward_stages:
                hook(g, b, e)

    def numerics_status(self, clear: bool = True) -> int:
        """OR of the numerics status words of the live plans (`Plan.status`; one device synchronisation per plan).
        Non-zero when a BatchNorm statistic, a BatchNorm-backward sum or a logit was not finite since the last
        clear — a diverged run, or a value outside what the precision mode represents: the default "split16" forward
        carries activations as fp16 (hi, lo) pairs and weights as fp16 pairs x 2^8, so |activation| >= 65520 or
        |weight| >= 256 overflows there where the reference's fp32 path does not ("fp32" mode has fp32 range)."""
        flags = getattr(self, "_evicted_status", 0)  # recorded by plans the LRU cache has dropped since
        if clear:
            self._evicted_status = 0
        for plan in list(self._plans.values()):
            flags |= plan.status(clear)
        return flags

    def check_numerics(self, clear: bool = True) -> None:
        """Raise FloatingPointError with what was recorded (see `numerics_status`); cheap enough for once per epoch.
        Under "16-mixed" a non-finite BACKWARD sum is the loss scaler's normal overflow probe (GradScaler skips that
        step and halves the scale) and is not reported."""
        ran_in = self._geom.precision  # the mode the flags were recorded in
        switched = self.guard_fp16_range()  # (a switch of mode drops the plans; set_precision keeps their status)
        flags = self.numerics_status(clear)
        if ran_in == "16-mixed":
            flags &= ~2
        if flags:
            what = "; ".join(msg for bit, msg in Plan.STATUS_BITS.items() if flags & bit)
            hint = ""
            if ran_in in ("split16", "16-mixed"):
                hint = (f" (precision {ran_in!r} carries fp16 operands: |activation| >= 65520 overflows"
                        + (f"; the BatchNorm parameters now allow that, the network continues in {self._geom.precision!r}" if switched
                           else "; set_precision('fp32') or 'bf16-mixed' has fp32 exponent range") + ")")
            raise FloatingPointError(f"MimoUNet: {what}{hint}")

    def mark_parameters_changed(self) -> None:
        """Call after writing the flat parameter / buffer storage through a raw pointer."""
        self._param_epoch += 1

    def _param_version(self) -> int:
        """Changes whenever the parameters / BatchNorm buffers may have changed.  The nn.Parameters are views
        of the flat storage but carry their OWN torch version counters (`load_state_dict`, `p.copy_()`, EMA
        swaps bump those, not the flat tensor's), so all of them are folded in; writers that go through
        `.data` or a raw pointer are invisible to torch and must call `mark_parameters_changed()`."""
        if self._flat_params is None:
            return 0
        v = self._flat_params._version + self._flat_buffers._version
        for t in self._versioned:
            v += t._version
        return (self._param_epoch << 40) + v + 1

    def _load_from_state_dict(self, *args, **kwargs):
        # any (partial) load may rewrite parameters in place: never serve packed weights derived before it
        r = super()._load_from_state_dict(*args, **kwargs)
        self._param_epoch += 1
        return r

    # flat views for the fused optimiser / gradient all-reduce
    def flat_parameters(self) -> torch.Tensor:
        return self._flat_params

    def flat_gradients(self) -> torch.Tensor:
        return self._flat_grads
