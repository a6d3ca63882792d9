from mimo_unet_amd.models.utils import (apply_input_transform, compute_uncertainties,  # noqa: F401
                                        flatten_subnetwork_dimension, repeat_subnetworks)
