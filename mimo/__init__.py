"""Import-path alias: `mimo.*` (the reference's package name) -> `mimo_unet_amd.*`,
so `scripts/train/*.py` and `scripts/test/*.py` of the reference run unchanged.

Only the hot path is mirrored here (`mimo.models.*`, `mimo.losses`, `mimo.metrics`, `mimo.utils`).
Everything else the reference's scripts import next (`mimo.tasks`, `mimo.datasets`, `mimo.callbacks`,
`mimo.regularization`, `mimo.visualization` — scripts/train/train_ndvi.py:10-13) resolves from the
reference's own tree: with this repo FIRST and a reference checkout SECOND on PYTHONPATH,
`pkgutil.extend_path` adds the reference's `mimo/` directory to this package's search path, so
sub-modules that do not exist here are found there, while `mimo.models` / `mimo.losses` stay ours."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)

from mimo_unet_amd import __version__  # noqa: E402,F401
