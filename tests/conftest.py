import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
