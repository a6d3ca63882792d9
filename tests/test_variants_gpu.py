"""Opt-in / fallback kernel variants selected by environment variables that the library reads once per
process: each group runs the convolution parity tests (and the golden training steps) in a child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# The switches are independent of each other, so they are exercised in three child processes instead of one per switch
# (a child costs ~10 s of interpreter / torch start-up on the GPU box); when a group fails, rerun its members one by one.
VARIANTS = {
    "fallback_paths": {
        "MIMO_CONV_WS": "0",               # non-specialised split convolution on every image size ...
        "MIMO_CONV_WIDE": "0",             # ... (the wide kernel would otherwise take its layers first)
        "MIMO_WGRAD_WS": "0",              # 4-wave weight gradient for 64x64 tiles too
        "MIMO_WGRAD_STREAM": "0",          # weight gradients on the caller's stream (the default is the side stream)
        "MIMO_HIP_GRAPH": "0",
        "MIMO_SKIP_IN_PLACE": "0",         # skip tensors copied into the concat buffers
        "MIMO_SKIP_GRAD_IN_PLACE": "0",    # skip-connection gradients copied out by fold_slice
        "MIMO_POOL_FUSED": "0",            # separate MaxPool2d pass after BatchNorm + ReLU
        "MIMO_FUSE_BN_IN": "0",            # the activation between the two convolutions of a block materialised
        "MIMO_FUSE_BWD_SRC": "0",          # pool_bwd / head_bwd as separate kernels writing the gradient tensors
        "MIMO_CONV_THIN": "0",             # the image convolution on the fp32 MFMA kernels / the split weight gradient
    },
    "specialised_kernels_plain": {
        "MIMO_CONV_WIDE": "0",             # 256-pixel kernels everywhere ...
        "MIMO_CONV_WS_MF2": "0",           # ... thin forward layers on 256-pixel tiles, one workgroup per CU
        "MIMO_CONV_WDMA": "0",             # convolution weights staged through registers, not by LDS-DMA
        "MIMO_CONV_PAIR_TAIL": "0",        # short last K chunks with one tap per MFMA
        "MIMO_WGRAD_SPLIT_MODE": "0",      # fixed split count of the weight gradient
    },
    "conv_wide_forced": {"MIMO_CONV_WIDE": "2"},  # every supported convolution on conv_wide.hip
    "image_conv_wgrad_forced": {"MIMO_CONV_THIN": "2"},  # 1-2-channel weight gradients on conv_thin.hip at every size
    # K split (conv3x3_ksplit) forced three-fold wherever a launch has whole 32-channel chunks (cfg1's 32 / 64 / 128-channel
    # layers; the 64- and 96-channel operator cases: a last split with one chunk)
    "conv_ksplit_forced": {"MIMO_CONV_KSPLIT": "3"},
    # the three-MFMA bf16-pair weight gradient of rounds 1-4 (the tests then apply its tighter bounds: tests/helpers.py)
    "wgrad_three_mfma": {"MIMO_WGRAD_NP": "3"},
}


def _run_variant(name):
    env = dict(os.environ, **VARIANTS[name])
    env.pop("MIMO_PARITY_LOG", None)  # the default-path run of the same tests writes the committed error log
    sel = ("tests/test_ops_gpu.py::test_conv3x3_forward_dgrad_wgrad "
           "tests/test_network_gpu.py::test_train_steps_match_reference_golden "
           "tests/test_network_gpu.py::test_mc_dropout_ensemble_golden "
           "tests/test_network_gpu.py::test_weights_beyond_the_old_fixed_fp16_scale_stay_finite_and_close "
           "tests/test_ops_gpu.py::test_storage_mode_conv_kernels_against_rounded_reference").split()
    return subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                           "-k", "split16 or mc_dropout or storage_mode or fixed_fp16_scale", *sel],
                          cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_kernel_variant(name):
    r = _run_variant(name)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
