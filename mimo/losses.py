from mimo_unet_amd.losses import *  # noqa: F401,F403
from mimo_unet_amd.losses import EvidentialLoss, GaussianNLL, LaplaceNLL, UncertaintyLoss  # noqa: F401
