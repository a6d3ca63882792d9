from mimo_unet_amd.models.mimo_components.loss_buffer import LossBuffer, softmax_temperature  # noqa: F401
