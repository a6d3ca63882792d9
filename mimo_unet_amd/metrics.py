"""r2 / mae / mse / rmse for the step logs (interface of mimo/metrics.py:22-34).  The
reference delegates to torchmetrics.functional, which is not a dependency here; these are
the same closed forms on flattened tensors."""
from typing import Dict, List, Optional

import torch


def _r2(y_hat, y):
    ss_res = torch.sum((y - y_hat) ** 2)
    ss_tot = torch.sum((y - y.mean()) ** 2)
    return 1 - ss_res / ss_tot


_METRICS = {
    "mae": lambda a, b: (a - b).abs().mean(),
    "mse": lambda a, b: ((a - b) ** 2).mean(),
    "rmse": lambda a, b: ((a - b) ** 2).mean().sqrt(),
    "r2": _r2,
    "mape": lambda a, b: ((a - b).abs() / b.abs().clamp_min(1.17e-06)).mean(),
}


def get_metric(metric: str):
    if metric not in _METRICS:
        raise ValueError(f"Unknown metric: {metric}")
    return _METRICS[metric]


def compute_regression_metrics(y_hat: torch.Tensor, y: torch.Tensor,
                               metrics: Optional[List[str]] = ("r2", "mae", "mse", "rmse")) -> Dict[str, torch.Tensor]:
    y, y_hat = y.detach(), y_hat.detach()
    return {m: get_metric(m)(y_hat, y) for m in metrics}
