from mimo_unet_amd.models.ensemble import EnsembleModule  # noqa: F401
