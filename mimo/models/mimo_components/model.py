from mimo_unet_amd.models.mimo_components.model import MimoUNet  # noqa: F401
