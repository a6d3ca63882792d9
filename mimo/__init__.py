"""Import-path alias: `mimo.*` (the reference's package name) -> `mimo_unet_amd.*`,
so `scripts/train/*.py` and `scripts/test/*.py` of the reference run unchanged."""
from mimo_unet_amd import __version__  # noqa: F401
