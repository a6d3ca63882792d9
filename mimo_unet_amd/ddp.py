"""Data-parallel plumbing for the MIMO U-Net step: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests).

The reference is single-GPU (`devices=1`, scripts/train/train_ndvi.py:67-76); the semantics
implemented are what Lightning DDP would do with its module: the batch is sharded, BatchNorm
statistics and the loss buffer stay per rank, gradients are averaged.  The engine keeps all
gradients in ONE flat buffer, so the exchange is a handful of bucketed all-reduces of slices of
that buffer (xGMI is per-link bound: few, large messages), issued asynchronously from inside the
backward — one slice per core block as it becomes final (RCCL runs them on its own stream) — and
waited for right before the optimiser."""
from __future__ import annotations

import contextlib
import os
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


def broadcast_buffers(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """BatchNorm running statistics under data parallelism.  Every rank normalises with the statistics of its OWN shard
    (no SyncBatchNorm — what Lightning DDP does with the reference module) and so keeps its own running_mean /
    running_var; parameters stay bit-identical across ranks, these buffers do not.  The checkpoint semantics are
    "rank 0's buffers" (Lightning saves on rank 0 only): before a checkpoint — and before any evaluation whose result
    must not depend on the rank — every rank takes rank `src`'s buffers.  MimoUnetModel / EvidentialUnetModel call this
    from `on_save_checkpoint` and `on_validation_epoch_start`."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    flat = getattr(module, "_flat_buffers", None)
    if flat is not None:
        dist.broadcast(flat, src, group=group)  # the engine keeps all of them in one flat tensor
        mark = getattr(module, "mark_parameters_changed", None)
        if mark is not None:
            mark()  # written through .data semantics: cached inference weights / eval constants are stale
        return
    for b in module.buffers():
        dist.broadcast(b.data, src, group=group)


class FlatGradientAllReducer:
    """Sum all-reduce of a flat gradient buffer, started range by range while the backward is still running;
    `scale` is what the optimiser must multiply gradients by afterwards (1/world: FlatAdam.reduce_scale).

    The engine announces ranges in backward order — heads + decoders, up3, up2, up1, down4, down3, down2,
    encoders — which walks the flat buffer (encoder | core | decoder | heads) from its tail to its head, so
    consecutive announcements are adjacent in memory.  Ranges are merged until `min_bucket_bytes` are pending
    (the 0.4 MB decoder and 1.3 MB up3 slices ride along with up2; xGMI collectives are latency-bound below a
    few MB) and split above `bucket_bytes`; cfg3 (60 MB of gradients) goes out as five collectives of 3.6-20.7 MB,
    the first of them after about a third of the backward."""

    ALGORITHMS = ("all_reduce", "reduce_scatter")

    def __init__(self, bucket_bytes: int = 64 << 20, group=None, min_bucket_bytes: int = 4 << 20,
                 algorithm: Optional[str] = None):
        """algorithm: "all_reduce" (default; MIMO_DDP_ALGO overrides) — one `dist.all_reduce` per bucket, RCCL picks the
        schedule — or "reduce_scatter": each bucket as an in-place reduce-scatter (rank r ends up with the sum of its
        1/W-th of the bucket) followed by an in-place all-gather, the direct form SURVEY §5 / §8(e) prefers on the fully
        connected 8-GPU xGMI topology (every link carries 1/W of the bucket twice instead of a ring pushing (W-1)/W of it
        through each link twice); the < W floats a bucket does not divide into go out as a small all-reduce.  Same sums on
        every rank either way (the two may differ from each other in the last bit for W > 2: the order of the W addends)."""
        self.group = group
        self.algorithm = algorithm or os.environ.get("MIMO_DDP_ALGO", "all_reduce")
        if self.algorithm not in self.ALGORITHMS:
            raise ValueError(f"FlatGradientAllReducer: algorithm {self.algorithm!r} not in {self.ALGORITHMS}")
        self.bucket_floats = max(1, bucket_bytes // 4)
        self.min_floats = max(1, min(min_bucket_bytes, bucket_bytes) // 4)
        self.world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        self.always = False  # issue the collectives even with one rank (functional check of the RCCL path)
        self.enabled = True  # False inside no_sync(): ranges announced by the backward are not reduced
        self._pending: List = []
        self._held = None  # (flat, begin, end): announced, not yet issued
        self.issued: List = []  # [(begin, end)] of the collectives of the current step (diagnostics / tests)
        self.last_issued: List = []  # ... of the step `finish()` completed last
        # (flat buffer, its torch version counter) right after the last collective issued on it: while the counter still
        # has that value nothing has written the gradients through torch since (no zero_grad(set_to_none=False), no
        # clipping), so a backward that ACCUMULATES into them would sum an already-reduced micro-batch over the ranks again
        self._reduced = None

    @property
    def scale(self) -> float:
        return 1.0 / self.world_size

    def _issue(self, flat: torch.Tensor, begin: int, end: int) -> None:
        for lo in range(begin, end, self.bucket_floats):
            hi = min(end, lo + self.bucket_floats)
            self.issued.append((lo, hi))
            if self.algorithm == "reduce_scatter" and (self.world_size > 1 or self.always):
                self._issue_reduce_scatter(flat, lo, hi)
            else:
                self._pending.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._reduced = (flat, flat._version)

    def _issue_reduce_scatter(self, flat: torch.Tensor, lo: int, hi: int) -> None:
        """flat[lo:hi] summed over the ranks as reduce-scatter + all-gather, both in place: rank r's output chunk is the
        r-th 1/W of the bucket itself (RCCL's in-place form: recvbuff == sendbuff + rank * count)."""
        W, r = self.world_size, dist.get_rank(self.group)
        chunk = (hi - lo) // W
        if chunk > 0:
            body = flat[lo: lo + W * chunk]
            mine = flat[lo + r * chunk: lo + (r + 1) * chunk]
            rs = dist.reduce_scatter_tensor(mine, body, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            if dist.get_backend(self.group) != "nccl":
                rs.wait()  # (RCCL orders the two on its stream; a host-side backend may run async works concurrently)
            else:
                self._pending.append(rs)
            self._pending.append(dist.all_gather_into_tensor(body, mine, group=self.group, async_op=True))
        if lo + W * chunk < hi:  # the remainder, fewer than W floats
            self._pending.append(dist.all_reduce(flat[lo + W * chunk: hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def reduced_version(self, flat: torch.Tensor):
        """torch version counter of `flat` after the last collective issued on it, None if there was none."""
        return self._reduced[1] if self._reduced is not None and self._reduced[0] is flat else None

    def _flush(self) -> None:
        if self._held is not None:
            flat, b, e = self._held
            self._held = None
            self._issue(flat, b, e)

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation (the counterpart of DistributedDataParallel.no_sync): run every micro-batch but the
        LAST one inside this context.  Their backward passes only accumulate locally; the last micro-batch's backward
        then announces — and all-reduces — the accumulated sum once.  Without it micro-batch 1 would be summed over the
        ranks when its ranges become final and again with micro-batch 2 (MimoUNet refuses that: it raises)."""
        prev, self.enabled = self.enabled, False
        try:
            yield self
        finally:
            self.enabled = prev

    @property
    def busy(self) -> bool:
        """collectives in flight or a range held for merging"""
        return bool(self._pending) or self._held is not None

    def start(self, flat: torch.Tensor, begin: int = 0, end: Optional[int] = None) -> None:
        """flat[begin:end] is final: issue its async all-reduce, or hold it to merge with an adjacent range."""
        if not self.enabled or (self.world_size == 1 and not self.always):
            return
        end = flat.numel() if end is None else end
        if self._held is not None:
            hf, hb, he = self._held
            if hf is flat and end == hb:
                begin, end = begin, he
            elif hf is flat and begin == he:
                begin, end = hb, end
            else:
                self._flush()
        self._held = (flat, begin, end)
        if end - begin >= self.min_floats:
            self._flush()

    def finish(self) -> None:
        self._flush()
        for w in self._pending:
            w.wait()
        if self._pending and self._reduced is not None:
            self._reduced = (self._reduced[0], self._reduced[0]._version)  # (a backend may bump the counter on completion)
        self._pending.clear()
        self.last_issued, self.issued = self.issued, []

    def all_reduce(self, flat: torch.Tensor) -> None:
        self.start(flat)
        self.finish()

    def attach(self, net) -> None:
        """Overlap with the backward: `net` (MimoUNet) calls `start` as soon as a range of its flat
        gradient buffer is final (after each backward stage, include/mimo_hip.h); call `finish()` before
        the optimiser step."""
        net.grad_ready_hook = self.start
        net.grad_sync = self  # lets the backward drain in-flight collectives before it rewrites the buffer, and see no_sync()
