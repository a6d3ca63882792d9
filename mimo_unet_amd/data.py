"""Host -> device side of the data path (SURVEY §8f "later" row).

The reference's loaders hand Lightning batches that are dicts of host tensors —
``{"image": [N,Ci,H,W] float32, "label": [N,Ct,H,W] float32}`` from
``mimo/datasets/nyuv2.py:38-53`` (plus ``"mask"`` where a dataset has one), collated by a
``DataLoader(pin_memory=True)`` (``mimo/tasks/sen12tp/sen12tp_datamodule.py:15-35``) — and
Lightning moves each batch to the GPU with one copy per tensor right before
``training_step``.  A copy from pageable memory on the step's own stream blocks the host
until that stream has drained (round 3 measured 2.4 % of the batch-32 step and 12 % at 4
images per GPU for one such copy), so the run-ahead the engine's asynchronous step depends
on is gone.

`DevicePrefetcher` wraps any iterable of such batches and yields the same dicts with the
tensors resident in HBM, keeping ``depth`` batches in flight:

* **worker thread** — drawing the next batch from the iterable (a ``DataLoader``'s
  ``__next__``) and issuing its uploads happen on a thread of their own, beside the thread
  that enqueues the training step: whatever blocks there — a copy from *pageable* memory is
  synchronous for its caller, the loader may be slow — blocks the worker, not the step;
* **copy stream** — the uploads run on a stream of their own, under the previous steps'
  kernels; pinned tensors are copied asynchronously in place, pageable ones by the
  driver's staged copy (measured 50 GB/s on the GPU boxes, profiles/r05/data_path.txt —
  a hand-rolled pinned staging ring filled with ``memmove`` reached 1-3 GB/s there, and
  ``Tensor.copy_`` between host tensors woke torch's whole intra-op thread pool per call);
* **device ring** — ``depth + 1`` sets of device tensors, allocated once per batch
  shape and owned by the prefetcher, not by one epoch's iterator: the next ``iter()``
  (the next epoch, or the loop after an early ``break``) reuses the same slots WITH their
  "consumed" events, so a slot the last steps of the previous epoch still read is not
  overwritten by the new epoch's first uploads (per-iterator slots were
  dropped at the end of an epoch with up to ``depth`` steps still pending, and the
  caching allocator could hand the same memory to the next epoch's slots at once);
  no allocator traffic per step;
* **event hand-off, resolved on the worker** — a slot is overwritten only after its
  "consumed" event (recorded when the consumer asks for the next batch, i.e. after
  everything that reads the slot has been enqueued) has completed, and is handed over
  only after its "uploaded" event has: the WORKER waits for both on the host, so neither
  GPU queue ever holds a wait on the other.  Measured (profiles/r05/data_path.txt §4):
  with the same two dependencies expressed as stream waits (``MIMO_PREFETCH_HANDOFF=gpu``)
  a pinned feed costs 0.25-0.3 ms per step at every batch size — 4.85 against 4.58 ms at
  4 images per GPU — although the copies themselves are the same 27 us SDMA transfers.
  The consuming thread only ever waits for the worker (counted in ``starved``), never
  for the GPU directly; through the worker it is held to ``depth`` steps ahead of the GPU.

The layout change NCHW -> NHWC (and the per-subnetwork gather) is not done here: the
engine's first kernel (`pack_input_kernel`) reads the NCHW batch as uploaded.

A yielded batch stays valid until the next one is drawn — the contract of a training loop
that uses each batch for one step."""
from __future__ import annotations

import os
import queue
import threading
from typing import Any, Dict, Iterable, Iterator, Optional

import torch


class _Slot:
    __slots__ = ("dev", "uploaded", "consumed", "spec", "keep")

    def __init__(self):
        self.dev: Dict[str, torch.Tensor] = {}
        self.uploaded: Optional[torch.cuda.Event] = None
        self.consumed: Optional[torch.cuda.Event] = None
        self.spec = None
        self.keep = None  # the host batch of the upload in flight (a loader's pinned tensors must outlive the copy)


_END, _STOP = object(), object()


class DevicePrefetcher:
    """Iterate `batches` (dicts of host tensors; non-tensor values pass through) as device-resident dicts.

    depth: batches uploaded ahead of the one being consumed (2 = double buffer in front of the step)."""

    def __init__(self, batches: Iterable[Dict[str, Any]], device="cuda", depth: int = 2):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("DevicePrefetcher uploads to an AMD GPU; got device " + str(device))
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.batches, self.depth = batches, int(depth)
        # high priority: not for the scheduling but for the hardware queue — streams of one priority class share a handful
        # of hardware queues, and a copy stream that lands on the training stream's queue is serialised with its kernels
        # (csrc/plan.hip, the side stream; profiles/r06/b4/queue_collision.txt); the priority classes use separate pools
        self.copy_stream = torch.cuda.Stream(self.device, priority=-1)
        # "host": the worker waits for the slot events itself; "gpu": stream waits (kept for the A/B, see the module text)
        self.handoff = os.environ.get("MIMO_PREFETCH_HANDOFF", "host")
        if self.handoff not in ("host", "gpu"):
            raise ValueError("MIMO_PREFETCH_HANDOFF must be 'host' or 'gpu'")
        self.starved = 0           # times the consumer found no uploaded batch ready and waited for the worker
        self.pageable_uploads = 0  # tensors that came from pageable memory (synchronous copies, on the worker)
        # the device ring and the worker of the iteration in progress live on the prefetcher: see "device ring" above
        self._slots = [_Slot() for _ in range(self.depth + 1)]
        self._worker: Optional[threading.Thread] = None
        self._stop_q: Optional["queue.Queue"] = None
        self._handed: Optional[_Slot] = None  # the slot of the batch the consumer holds right now

    def __len__(self) -> int:
        return len(self.batches)  # type: ignore[arg-type]

    # -- one upload (worker thread) ---------------------------------------------------------------------------------
    def _upload(self, slot: _Slot, batch: Dict[str, Any]) -> Dict[str, Any]:
        spec = tuple((k, tuple(v.shape), v.dtype) for k, v in batch.items() if torch.is_tensor(v))
        out: Dict[str, Any] = {}
        with torch.cuda.stream(self.copy_stream):
            if slot.consumed is not None:
                # the step that read this slot's device tensors is done
                if self.handoff == "host":
                    slot.consumed.synchronize()
                else:
                    self.copy_stream.wait_event(slot.consumed)
            if slot.spec != spec:  # first use, or a ragged last batch: (re)allocate this slot's buffers
                # (behind the wait above: the old buffers go back to the allocator only once their last reader is done;
                # allocated under the copy stream's context: the caching allocator orders reuse of the memory on it)
                slot.dev = {k: torch.empty(shape, dtype=dt, device=self.device) for k, shape, dt in spec}
                slot.spec = spec
            for k, v in batch.items():
                if not torch.is_tensor(v):
                    out[k] = v
                    continue
                if not v.is_cuda and not v.is_pinned():
                    self.pageable_uploads += 1
                slot.dev[k].copy_(v, non_blocking=True)
                out[k] = slot.dev[k]
            slot.keep = batch
            slot.uploaded = torch.cuda.Event()
            slot.uploaded.record(self.copy_stream)
            if self.handoff == "host":
                slot.uploaded.synchronize()
        return out

    def _produce(self, it: Iterator[Dict[str, Any]], free_q: "queue.Queue", ready_q: "queue.Queue") -> None:
        try:
            torch.cuda.set_device(self.device)
            for batch in it:
                slot = free_q.get()
                if slot is _STOP:
                    return
                ready_q.put((slot, self._upload(slot, batch)))
            ready_q.put(_END)
        except BaseException as e:  # surfaces in the consuming thread
            ready_q.put(e)

    def _retire_worker(self) -> None:
        """The previous iteration's worker is gone before its slots are handed out again (it may be inside an upload)."""
        if self._worker is not None:
            if self._stop_q is not None:
                self._stop_q.put(_STOP)
            self._worker.join()
            self._worker = self._stop_q = None

    def _release(self, slot: _Slot) -> None:
        # everything that reads this slot's batch has been enqueued by now (the caller came back for more, or left)
        slot.consumed = torch.cuda.Event()
        slot.consumed.record(torch.cuda.current_stream(self.device))
        if self._handed is slot:
            self._handed = None

    def __iter__(self) -> Iterator[Dict[str, Any]]:
        self._retire_worker()
        if self._handed is not None:
            # an iterator that was abandoned without being closed (its `finally` has not run yet): the batch it handed
            # out last is over by the contract above — "valid until the next one is drawn"
            self._release(self._handed)
        free_q: "queue.Queue" = queue.Queue()
        ready_q: "queue.Queue" = queue.Queue()
        for slot in self._slots:  # with the "consumed" events the previous iteration left on them
            free_q.put(slot)
        worker = threading.Thread(target=self._produce, args=(iter(self.batches), free_q, ready_q), daemon=True,
                                  name="DevicePrefetcher")
        self._worker, self._stop_q = worker, free_q
        worker.start()
        previous: Optional[_Slot] = None
        try:
            while True:
                if previous is not None:
                    self._release(previous)
                    free_q.put(previous)
                    previous = None
                try:
                    item = ready_q.get_nowait()
                except queue.Empty:
                    self.starved += 1
                    item = ready_q.get()
                if item is _END:
                    return
                if isinstance(item, BaseException):
                    raise item
                slot, dev_batch = item
                if self.handoff != "host":  # ("host": the worker handed the slot over after the upload had completed)
                    torch.cuda.current_stream(self.device).wait_event(slot.uploaded)
                previous = self._handed = slot
                yield dev_batch
        finally:
            if previous is not None and self._stop_q is free_q:
                # the last batch handed out (end of the epoch, an early break, an exception in the loop body): the steps
                # that read it may still be pending when the next iteration's worker reaches this slot.  (Not when a newer
                # iteration already owns the ring: it released this slot when it started.)
                self._release(previous)
            free_q.put(_STOP)  # a consumer that stops early leaves no worker waiting for a slot
