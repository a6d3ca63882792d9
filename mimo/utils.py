from mimo_unet_amd.utils import count_trainable_parameters, dir_path  # noqa: F401
