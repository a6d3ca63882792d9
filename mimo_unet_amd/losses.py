"""Probabilistic regression losses with the reference's interface
(``mimo/losses.py:4-192``: ``UncertaintyLoss``, ``GaussianNLL``, ``LaplaceNLL``).

On the training hot path the loss is evaluated inside libmimo_hip.so
(`mimo_loss_forward` / the head backward kernel); these classes are the API surface the
callers hold (`model.loss_fn`, `EnsembleModule.loss_fn`) and serve the off-path uses
(validation's combined NLL, FGSM evaluation) with plain tensor arithmetic."""
from abc import ABC, abstractmethod

import torch


class UncertaintyLoss(torch.nn.Module, ABC):
    num_distribution_params = 2
    name = ""

    def __init__(self, eps_min: float = 1e-5, eps_max: float = 1e3) -> None:
        super().__init__()
        self.eps_min = eps_min
        self.eps_max = eps_max

    @abstractmethod
    def _nll(self, diff: torch.Tensor, param: torch.Tensor) -> torch.Tensor:
        ...

    @abstractmethod
    def std(self, mu: torch.Tensor, log_param: torch.Tensor) -> torch.Tensor:
        ...

    @abstractmethod
    def _param_from_std(self, std: torch.Tensor) -> torch.Tensor:
        ...

    def _value_clamped(self, t: torch.Tensor) -> torch.Tensor:
        # the clamp changes the value only; gradients flow as if it were absent (losses.py:66-68,153-155)
        t = t.clone()
        with torch.no_grad():
            t.clamp_(min=self.eps_min, max=self.eps_max)
        return t

    def forward(self, y_hat, log_param, y, *, mask=None, reduce_mean: bool = True):
        loss = self._nll(y_hat - y, self._value_clamped(torch.exp(log_param)))
        if mask is not None:
            loss = loss * mask
        return torch.mean(loss) if reduce_mean else loss

    def mode(self, mu, log_param):
        return mu

    def calculate_dist_param(self, std: torch.Tensor, *, log: bool = False) -> torch.Tensor:
        param = self._value_clamped(self._param_from_std(std))
        return torch.log(param) if log else param

    @classmethod
    def from_name(cls, name: str) -> "UncertaintyLoss":
        if name == "gaussian_nll":
            return GaussianNLL()
        if name == "laplace_nll":
            return LaplaceNLL()
        raise ValueError(f"Unknown loss function: {name}")


class GaussianNLL(UncertaintyLoss):
    name = "gaussian_nll"

    def _nll(self, diff, variance):
        return torch.log(variance) + diff ** 2 / variance

    def std(self, mu, log_variance):
        return torch.exp(log_variance) ** 0.5

    def _param_from_std(self, std):
        return std ** 2


class LaplaceNLL(UncertaintyLoss):
    name = "laplace_nll"

    def _nll(self, diff, scale):
        return torch.log(scale) + diff.abs() / scale

    def std(self, mu, log_scale):
        return torch.exp(log_scale) * (2 ** 0.5)

    def _param_from_std(self, std):
        return std / (2 ** 0.5)


class EvidentialLoss(torch.nn.Module):
    """Deep-evidential-regression loss on Normal-Inverse-Gamma outputs, interface of the reference's
    ``EvidentialLoss`` (``mimo/losses.py:195-271``): ``evidential_output`` is ``[B, 4, H, W]`` =
    (gamma, v, alpha, beta) with v, beta > 0 and alpha > 1.

        L = Gamma(alpha - 1/2) / (4 Gamma(alpha) v sqrt(beta)) * (2 beta (1 + v) + (2 alpha - 1) v (y - gamma)^2)
            + coeff-free regulariser (y - gamma)^2 (2 alpha + v)

    Evaluated with tensor arithmetic on the device behind the HIP backbone (element-wise on a
    [B,4,H,W] tensor: off the convolution hot path)."""
    num_distribution_params = 4

    def __init__(self, coeff: float = 1.0) -> None:
        super().__init__()
        self.coeff = coeff

    @staticmethod
    def evidential_loss(mu, v, alpha, beta, targets):
        sq = (targets - mu) ** 2
        gamma_ratio = torch.exp(torch.lgamma(alpha - 0.5)) / (4.0 * torch.exp(torch.lgamma(alpha)) * v * torch.sqrt(beta))
        sos = gamma_ratio * (2.0 * beta * (1.0 + v) + (2.0 * alpha - 1.0) * v * sq)
        reg = sq * (2.0 * alpha + v)
        return sos + reg

    def forward(self, evidential_output, y_true, *, mask=None, reduce_mean: bool = False) -> torch.Tensor:
        gamma, v, alpha, beta = torch.unbind(evidential_output, dim=1)
        loss = self.evidential_loss(gamma, v, alpha, beta, y_true.squeeze(dim=1))
        if mask is not None:
            loss = loss * mask
        return torch.mean(loss) if reduce_mean else loss

    @staticmethod
    def mode(evidential_output):
        return evidential_output[:, 0]

    @staticmethod
    def aleatoric_var(evidential_output):
        _, _, alpha, beta = torch.unbind(evidential_output, dim=1)
        return beta / (alpha - 1.0)

    @staticmethod
    def epistemic_var(evidential_output):
        _, v, alpha, beta = torch.unbind(evidential_output, dim=1)
        return beta / (v * (alpha - 1.0))
