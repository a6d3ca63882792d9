"""Host-side loss ring buffer -> softmax-temperature subnetwork weights.

Mirrors the interface of the reference's
``mimo/models/mimo_components/loss_buffer.py:3-74`` (``LossBuffer``,
``softmax_temperature``).  [S]-sized arithmetic.  Unlike the reference's CPU tensor, the ring follows
the device of the losses it is fed, so a training step needs no device-to-host copy (and no host sync)
between its forward and its backward."""
import torch


def softmax_temperature(x: torch.Tensor, temperature: float = 1.0) -> torch.Tensor:
    if not temperature > 0:
        raise AssertionError("Temperature should be positive.")
    return torch.softmax(x / temperature, dim=-1)


class LossBuffer:
    """Ring of the last ``buffer_size`` per-subnetwork losses (zero-initialised; the mean
    is taken over ALL rows, filled or not — loss_buffer.py:54-63)."""

    def __init__(self, subnetworks: int, temperature: float, buffer_size: int) -> None:
        self.index = 0
        self.temperature = temperature
        self.buffer_size = buffer_size
        self.subnetworks = subnetworks
        self.buffer = torch.zeros(buffer_size, subnetworks)

    def add(self, loss: torch.Tensor) -> None:
        if self.buffer_size == 0:
            return
        if self.buffer.device != loss.device:
            self.buffer = self.buffer.to(loss.device)  # once: the ring lives where the losses are produced
        self.buffer[self.index] = loss.detach()
        self.index = (self.index + 1) % self.buffer_size

    def get_mean(self) -> torch.Tensor:
        if self.buffer_size == 0:
            return torch.zeros(self.subnetworks, device=self.buffer.device)
        return self.buffer.mean(dim=0)

    def get_weights(self) -> torch.Tensor:
        mean = self.get_mean()
        return softmax_temperature(mean, temperature=self.temperature) * len(mean)
