"""ctypes binding of libmimo_hip.so (C ABI declared in include/mimo_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  If the
shared object is missing, importing the binding raises with build instructions.
"""
from __future__ import annotations

import ctypes as C
import os

# torch must be imported before libmimo_hip.so is dlopen'ed: the PyTorch-ROCm wheel bundles its own
# libamdhip64.so.7; loading ours first would bind the process to /opt/rocm's copy (same SONAME) and
# leave torch and the engine on mismatched HIP runtimes ("no ROCm-capable device is detected").
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIMO_HIP_LIB: another build of the same library (A/B runs of two builds on one GPU box)
LIB_PATH = os.environ.get("MIMO_HIP_LIB") or os.path.join(_HERE, "libmimo_hip.so")


class MimoHipError(RuntimeError):
    pass


class MimoConfig(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32), ("out_channels", C.c_int32), ("num_subnetworks", C.c_int32),
        ("filter_base_count", C.c_int32), ("batch", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
        ("encoder_dropout_rate", C.c_float), ("core_dropout_rate", C.c_float), ("decoder_dropout_rate", C.c_float),
        ("bn_eps", C.c_float), ("bn_momentum", C.c_float), ("loss_kind", C.c_int32),
        ("eps_min", C.c_float), ("eps_max", C.c_float), ("device", C.c_int32), ("precision", C.c_int32),
        ("inference_only", C.c_int32), ("center_dropout_rate", C.c_float), ("final_dropout_rate", C.c_float),
        ("norm_kind", C.c_int32), ("act_kind", C.c_int32), ("up_kind", C.c_int32),
    ]


class ForwardArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("stride_n", C.c_int64), ("stride_s", C.c_int64), ("perm", C.c_void_p),
        ("training", C.c_int32), ("drop_masks", C.POINTER(C.c_void_p)), ("out", C.c_void_p),
        ("elem_masks", C.POINTER(C.c_void_p)), ("no_grad", C.c_int32), ("param_version", C.c_int64),
        ("rng_sites", C.POINTER(C.c_uint8)), ("rng_seed", C.c_uint64), ("rng_offset", C.c_uint64),
        ("x_rows", C.c_int64),
    ]


LOSS_KINDS = {"laplace_nll": 0, "gaussian_nll": 1}
# "bf16-mixed" / "16-mixed": Lightning's names for bf16 / fp16 autocast training — here 16-bit storage + operands
PRECISIONS = {"fp32": 0, "split16": 1, "bf16": 2, "bf16-mixed": 3, "16-mixed": 4}

_lib = None

_P, _I, _L, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_SIGNATURES = {
    "mimo_last_error": (C.c_char_p, []),
    "mimo_version": (C.c_int, []),
    "mimo_plan_create": (C.c_int, [C.POINTER(MimoConfig), C.POINTER(_P)]),
    "mimo_plan_destroy": (None, [_P]),
    "mimo_plan_workspace_bytes": (C.c_size_t, [_P]),
    "mimo_plan_num_tensors": (C.c_int, [_P]),
    "mimo_plan_tensor_info": (C.c_int, [_P, C.c_int, C.c_char_p, C.c_int, C.POINTER(_L), C.POINTER(C.c_int),
                                        C.POINTER(C.c_int), C.POINTER(_L)]),
    "mimo_plan_param_floats": (_L, [_P]),
    "mimo_plan_buffer_floats": (_L, [_P]),
    "mimo_plan_bind": (C.c_int, [_P, _P, _P, _P]),
    "mimo_plan_dropout_mask": (C.c_int, [_P, C.c_int, _P, _P]),
    "mimo_plan_status": (C.c_int, [_P, C.POINTER(C.c_int32), C.c_int32, _P]),
    "mimo_plan_num_double_convs": (C.c_int, [_P]),
    "mimo_plan_double_conv_channels": (C.c_int, [_P, C.c_int]),
    "mimo_forward": (C.c_int, [_P, C.POINTER(ForwardArgs), _P]),
    "mimo_loss_forward": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "mimo_backward": (C.c_int, [_P, _P, _P, _P, _P]),
    "mimo_backward_stage": (C.c_int, [_P, C.c_int, _P, _P, _P, _P]),
    "mimo_backward_stage_async": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, C.POINTER(C.c_void_p)]),
    "mimo_plan_encoder_param_floats": (_L, [_P]),
    "mimo_plan_num_backward_stages": (C.c_int, [_P]),
    "mimo_plan_backward_stage_range": (C.c_int, [_P, C.c_int, C.POINTER(_L), C.POINTER(_L)]),
    "mimo_plan_profile": (C.c_int, [_P, C.c_int]),
    "mimo_plan_profile_read": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(_L), C.POINTER(C.c_double),
                                         C.POINTER(C.c_double)]),
    "mimo_plan_profile_read_tier": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mimo_plan_profile_read_kind_tier": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "mimo_adam_step": (C.c_int, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P]),
    "mimo_adam_step_amp": (C.c_int, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _P, _F, _P, _P, _P]),
    "mimo_uncertainties": (C.c_int, [_P, _P, _I, _I, _I, _L, _I, _P, _P, _P, _P]),
    "mimo_evidential_forward": (C.c_int, [_P, _P, _P, _I, _L, _P, _P, _P]),
    "mimo_evidential_backward": (C.c_int, [_P, _P, _P, _P, _P, _I, _L, _P, _P]),
    "mimo_validation_epilogue": (C.c_int, [_P, _P, _P, _I, _I, _I, _L, _I, _F, _F, _P, _P, _P, _P, _P, _P, _I, _P]),
    "mimo_loss_buffer_step": (C.c_int, [_P, _I, _I, _I, _F, _P, _P, _P, _P, _P]),
    "mimo_training_epilogue": (C.c_int, [_P, _P, _P, _I, _I, _I, _L, _I, _P, _P, _P, _P, _P, _P, _I, _P]),
    "mimo_op_conv3x3_forward": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mimo_op_conv3x3_dgrad": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mimo_op_conv3x3_wgrad": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mimo_op_maxpool2x2": (C.c_int, [_P, _P, _I, _I, _I, _I, _P]),
    "mimo_op_upsample_cat": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def load():
    """Load the shared library (once) and attach argument/return types."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MimoHipError(
            f"{LIB_PATH} not found: build it with `make` (hipcc --offload-arch=gfx950) or "
            "`python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().mimo_last_error()
        raise MimoHipError(f"{what or 'libmimo_hip'} failed (status {rc}): {msg.decode() if msg else '?'}")


def ptr(t) -> int:
    """Raw device pointer of a torch tensor (or 0 for None)."""
    return 0 if t is None else t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
