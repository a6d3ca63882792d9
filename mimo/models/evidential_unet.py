from mimo_unet_amd.models.evidential_unet import EvidentialUnetModel  # noqa: F401
