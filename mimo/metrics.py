from mimo_unet_amd.metrics import compute_regression_metrics, get_metric  # noqa: F401
