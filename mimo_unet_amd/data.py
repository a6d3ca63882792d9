"""Host -> device side of the data path (SURVEY §8f "later" row).

The reference's loaders hand Lightning batches that are dicts of host tensors —
``{"image": [N,Ci,H,W] float32, "label": [N,Ct,H,W] float32}`` from
``mimo/datasets/nyuv2.py:38-53`` (plus ``"mask"`` where a dataset has one), collated by a
``DataLoader(pin_memory=True)`` (``mimo/tasks/sen12tp/sen12tp_datamodule.py:15-35``) — and
Lightning moves each batch to the GPU with one copy per tensor right before
``training_step``.  A pageable copy blocks the host until the stream has drained (round 3
measured 2.4 % of the batch-32 step and 12 % at 4 images per GPU for one such copy), so
the run-ahead the engine's asynchronous step depends on is gone.

`DevicePrefetcher` wraps any iterable of such batches and yields the same dicts with the
tensors resident in HBM, keeping ``depth`` batches in flight:

* **pinned staging ring** — a batch whose tensors are not already pinned is copied into
  a page-locked slot (allocated once per shape); pinned batches are uploaded in place;
* **copy stream** — the uploads are ``non_blocking`` copies on a stream of their own,
  so they run under the previous step's kernels;
* **device ring** — ``depth + 1`` sets of device tensors, allocated once; no allocator
  traffic per step;
* **event hand-off** — the consumer's stream waits on the slot's "uploaded" event, and
  the copy stream waits on the slot's "consumed" event (recorded when the consumer asks
  for the next batch, i.e. after everything that reads the slot has been enqueued)
  before the slot is overwritten.  The host never waits for the GPU unless it is more
  than ``depth`` batches ahead of the uploads.

The layout change NCHW -> NHWC (and the per-subnetwork gather) is not done here: the
engine's first kernel (`pack_input_kernel`) reads the NCHW batch as uploaded.

A yielded batch stays valid until ``depth`` further batches have been drawn — the
contract of a training loop that uses each batch for one step."""
from __future__ import annotations

from typing import Any, Dict, Iterable, Iterator, List, Optional

import torch


def _host_copy(dst: torch.Tensor, src: torch.Tensor) -> None:
    """pageable -> pinned staging copy on the calling thread.  Not `dst.copy_(src)`: torch parallelises large host copies
    over its intra-op thread pool, and waking that pool once per tensor and step from an otherwise GPU-bound loop cost
    5-20 ms per step on the 256-core GPU boxes (measured: cfg3 at 4 images per GPU 4.6 -> 24 ms per step,
    profiles/r05/data_path.txt); one memmove of a contiguous tensor is 0.3 ms per 3 MB and releases the GIL."""
    if src.is_contiguous() and dst.is_contiguous() and src.dtype == dst.dtype:
        import ctypes
        ctypes.memmove(dst.data_ptr(), src.data_ptr(), src.numel() * src.element_size())
    else:
        dst.copy_(src)


class _Slot:
    __slots__ = ("host", "dev", "uploaded", "consumed", "spec")

    def __init__(self):
        self.host: Dict[str, torch.Tensor] = {}
        self.dev: Dict[str, torch.Tensor] = {}
        self.uploaded: Optional[torch.cuda.Event] = None
        self.consumed: Optional[torch.cuda.Event] = None
        self.spec = None


class DevicePrefetcher:
    """Iterate `batches` (dicts of host tensors; non-tensor values pass through) as device-resident dicts.

    depth: batches uploaded ahead of the one being consumed (2 = double buffer in front of the step).
    pinned: None = look at the first batch (`Tensor.is_pinned()` per tensor) and assume the loader keeps doing what it
    did; True / False = the caller states what the loader yields (`DataLoader(pin_memory=...)`).  A wrong assumption is
    slow, not wrong: a pageable tensor taken for pinned is uploaded by torch's blocking copy, a pinned one taken for
    pageable goes through the staging ring."""

    def __init__(self, batches: Iterable[Dict[str, Any]], device="cuda", depth: int = 2, pinned: Optional[bool] = None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("DevicePrefetcher uploads to an AMD GPU; got device " + str(device))
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.batches, self.depth = batches, int(depth)
        self.copy_stream = torch.cuda.Stream(self.device)
        self._slots: List[_Slot] = [_Slot() for _ in range(self.depth + 1)]
        self._pinned: Dict[str, bool] = {}
        self._pinned_default = pinned
        # times the host waited for an upload slot still in flight: it was `depth` batches ahead of the copy stream (the
        # bounded run-ahead working as designed when the GPU is the bottleneck; never inside the first `depth` batches)
        self.throttle_waits = 0
        self.staged_copies = 0   # host tensors that were pageable and went through the pinned ring

    def __len__(self) -> int:
        return len(self.batches)  # type: ignore[arg-type]

    def _is_pinned(self, key: str, v: torch.Tensor) -> bool:
        if self._pinned_default is not None:
            return self._pinned_default
        known = self._pinned.get(key)
        if known is None:
            known = self._pinned[key] = bool(v.is_pinned())
        return known

    # -- one upload -------------------------------------------------------------------------------------------------
    def _upload(self, slot: _Slot, batch: Dict[str, Any]) -> Dict[str, Any]:
        spec = tuple((k, tuple(v.shape), v.dtype) for k, v in batch.items() if torch.is_tensor(v))
        if slot.spec != spec:  # first use, or a ragged last batch: (re)allocate this slot's buffers
            slot.host = {}
            slot.dev = {k: torch.empty(shape, dtype=dt, device=self.device) for k, shape, dt in spec}
            slot.spec = spec
            # the allocator may hand out memory that kernels already enqueued on the consumer's stream still read:
            # the first upload into it is ordered behind them
            fresh = torch.cuda.Event()
            fresh.record(torch.cuda.current_stream(self.device))
            self.copy_stream.wait_event(fresh)
        if slot.uploaded is not None and not slot.uploaded.query():
            # the staging buffer of this slot is still being read by its previous upload: the host is more than
            # `depth` batches ahead of the copy stream
            self.throttle_waits += 1
            slot.uploaded.synchronize()
        out: Dict[str, Any] = {}
        with torch.cuda.stream(self.copy_stream):
            if slot.consumed is not None:
                self.copy_stream.wait_event(slot.consumed)  # the step that read this slot's device tensors is done
            for k, v in batch.items():
                if not torch.is_tensor(v):
                    out[k] = v
                    continue
                if v.is_cuda:
                    slot.dev[k].copy_(v, non_blocking=True)
                elif self._is_pinned(k, v) and v.is_contiguous():
                    slot.dev[k].copy_(v, non_blocking=True)
                    slot.host[k] = v  # keep the loader's pinned tensor alive until the copy has run
                else:
                    h = slot.host.get(k)
                    if h is None or h.shape != v.shape or h.dtype != v.dtype or h is v:
                        h = torch.empty(v.shape, dtype=v.dtype).pin_memory()
                    _host_copy(h, v)
                    slot.host[k] = h
                    self.staged_copies += 1
                    slot.dev[k].copy_(h, non_blocking=True)
                out[k] = slot.dev[k]
            slot.uploaded = torch.cuda.Event()
            slot.uploaded.record(self.copy_stream)
        return out

    def __iter__(self) -> Iterator[Dict[str, Any]]:
        it = iter(self.batches)
        ring: List[Any] = []  # (slot, device batch) uploaded and not yet yielded, oldest first
        nslot = 0
        previous: Optional[_Slot] = None

        def fill():
            nonlocal nslot
            while len(ring) < self.depth:
                try:
                    b = next(it)
                except StopIteration:
                    return
                slot = self._slots[nslot % len(self._slots)]
                nslot += 1
                ring.append((slot, self._upload(slot, b)))

        fill()
        while ring:
            consumer = torch.cuda.current_stream(self.device)
            if previous is not None:
                # everything that reads the previous batch has been enqueued by now (the caller came back for more)
                previous.consumed = torch.cuda.Event()
                previous.consumed.record(consumer)
            slot, dev_batch = ring.pop(0)
            consumer.wait_event(slot.uploaded)
            previous = slot
            fill()
            yield dev_batch
        if previous is not None:
            previous.consumed = torch.cuda.Event()
            previous.consumed.record(torch.cuda.current_stream(self.device))
