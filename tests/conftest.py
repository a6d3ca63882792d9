import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # the CPU oracle on a many-core host: torch's default (one thread per core) is SLOWER than 16 threads on the
    # 256-core GPU boxes (tests/tools/cpu_thread_scan.py: 16 thr 3.4, 64 thr 1.7 images/s) — the full-resolution
    # configuration tests spend their time there
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def built_library():
    """libmimo_hip.so next to the package; built with `make` (hipcc cross-compiles gfx950 without a GPU) when a
    fresh checkout has not been through __graft_entry__.build() yet."""
    so = os.path.join(ROOT, "mimo_unet_amd", "libmimo_hip.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.run(["make", "-j4"], cwd=ROOT, check=True, capture_output=True)
    return so
