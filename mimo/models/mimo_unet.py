from mimo_unet_amd.models.mimo_unet import MimoUnetModel  # noqa: F401
