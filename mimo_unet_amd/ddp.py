"""Data-parallel plumbing for the MIMO U-Net step: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).

The reference is single-GPU (`devices=1`, scripts/train/train_ndvi.py:67-76); the semantics
implemented are what Lightning DDP would do with its module: the batch is sharded, BatchNorm
statistics and the loss buffer stay per rank, gradients are averaged.  The engine keeps all
gradients in ONE flat buffer, so the exchange is a handful of large bucketed all-reduces of
slices of that buffer (xGMI is per-link bound: few, large messages), issued asynchronously
(RCCL runs them on its own stream) and waited for right before the optimiser."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist


def shard_batch(batch: Dict[str, Optional[torch.Tensor]], rank: int, world_size: int) -> Dict[str, Optional[torch.Tensor]]:
    """Rank r takes images [r*N/W, (r+1)*N/W) of every tensor in the batch dict."""
    out = {}
    for k, v in batch.items():
        if v is None:
            out[k] = None
            continue
        n = v.shape[0]
        if n % world_size:
            raise ValueError(f"batch of {n} does not split over {world_size} ranks")
        per = n // world_size
        out[k] = v[rank * per:(rank + 1) * per]
    return out


class FlatGradientAllReducer:
    """Sum all-reduce of a flat gradient buffer in buckets; `scale` is what the optimiser must
    multiply gradients by afterwards (1/world: FlatAdam.grad_scale)."""

    def __init__(self, bucket_bytes: int = 64 << 20, group=None):
        self.group = group
        self.bucket_floats = max(1, bucket_bytes // 4)
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.always = False  # issue the collectives even with one rank (functional check of the RCCL path)
        self._pending: List = []

    @property
    def scale(self) -> float:
        return 1.0 / self.world_size

    def start(self, flat: torch.Tensor, begin: int = 0, end: Optional[int] = None) -> None:
        """Issue async all-reduces for flat[begin:end] (bucketed)."""
        if self.world_size == 1 and not self.always:
            return
        end = flat.numel() if end is None else end
        for lo in range(begin, end, self.bucket_floats):
            hi = min(end, lo + self.bucket_floats)
            self._pending.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self) -> None:
        for w in self._pending:
            w.wait()
        self._pending.clear()

    def all_reduce(self, flat: torch.Tensor) -> None:
        self.start(flat)
        self.finish()

    def attach(self, net) -> None:
        """Overlap with the backward: `net` (MimoUNet) calls `start` as soon as a range of its flat
        gradient buffer is final (core + decoder after backward stage 0, encoders after stage 1);
        call `finish()` before the optimiser step."""
        net.grad_ready_hook = self.start
