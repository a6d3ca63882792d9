"""Opt-in / fallback kernel variants selected by environment variables that the library reads once per
process: each runs the convolution parity tests (and one golden training step) in a child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VARIANTS = {
    "conv_ws_off": {"MIMO_CONV_WS": "0"},               # non-specialised split convolution on every image size
    "wgrad_ws_off": {"MIMO_WGRAD_WS": "0"},             # 4-wave weight gradient for 64x64 tiles too
    "wgrad_split_mode0": {"MIMO_WGRAD_SPLIT_MODE": "0"},
    "side_stream_off": {"MIMO_WGRAD_STREAM": "0"},      # weight gradients on the caller's stream (the default is the side stream)
    "no_graph": {"MIMO_HIP_GRAPH": "0"},
    "skip_copy": {"MIMO_SKIP_IN_PLACE": "0"},           # skip tensors copied into the concat buffers
    "skip_grad_copy": {"MIMO_SKIP_GRAD_IN_PLACE": "0"},  # skip-connection gradients copied out by fold_slice
    "pool_fused_off": {"MIMO_POOL_FUSED": "0"},         # separate MaxPool2d pass after BatchNorm + ReLU
    "conv_ws_mf2_off": {"MIMO_CONV_WS_MF2": "0"},       # thin forward layers on 256-pixel tiles, one workgroup per CU
    "conv_wdma_off": {"MIMO_CONV_WDMA": "0"},           # convolution weights staged through registers, not by LDS-DMA
    "conv_pair_tail_off": {"MIMO_CONV_PAIR_TAIL": "0"},  # short last K chunks with one tap per MFMA
    "conv_wide_forced": {"MIMO_CONV_WIDE": "2"},        # every supported split16 convolution on conv_wide.hip
    "conv_wide_off": {"MIMO_CONV_WIDE": "0"},           # ... and none of them (256-pixel kernels everywhere)
}


def _run_variant(name):
    env = dict(os.environ, **VARIANTS[name])
    env.pop("MIMO_PARITY_LOG", None)  # the default-path run of the same tests writes the committed error log
    sel = ("tests/test_ops_gpu.py::test_conv3x3_forward_dgrad_wgrad "
           "tests/test_network_gpu.py::test_train_steps_match_reference_golden "
           "tests/test_network_gpu.py::test_mc_dropout_ensemble_golden").split()
    if name.startswith("conv_wide"):  # the wide kernel also serves the 16-bit storage modes
        sel.append("tests/test_ops_gpu.py::test_storage_mode_conv_kernels_against_rounded_reference")
    return subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                           "-k", "split16 or mc_dropout or storage_mode", *sel],
                          cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)


@pytest.fixture(scope="module")
def variant_runs():
    """All variants start when the first one is asked for, four child processes at a time (they share the GPU; most of a
    child's time is the interpreter and torch starting up), so the module costs about a quarter of the serial time."""
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=4)
    futures = {name: pool.submit(_run_variant, name) for name in sorted(VARIANTS)}
    yield futures
    pool.shutdown(wait=True)


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_kernel_variant(name, variant_runs):
    r = variant_runs[name].result()
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
