"""MIMO batch helpers with the reference's names (``mimo/models/utils.py:5-101``).

On the hot path the gather of `apply_input_transform` is fused into the first kernel of
libmimo_hip.so (``perm`` argument of ``mimo_forward``); `draw_subnetwork_permutations`
produces exactly the index tensor the reference draws.  The functions below keep the
reference's call signatures for callers that want materialised tensors."""
from typing import Optional

import torch


_PINNED_RING = {}   # (k, device) -> {"slots": [[pinned int64 buffer, event of its last upload]], "next": int}
_PINNED_DEPTH = 16  # uploads in flight per (k, device) before a draw waits for the oldest one


def _cpu_randperms_on_device(k: int, count: int, device) -> torch.Tensor:
    """`count` consecutive `torch.randperm(k)` draws from the CPU generator as one [count, k] tensor on `device` — the
    reference draws the per-subnetwork shuffles there, one after the other (utils.py:31-34), and the parity tests replay
    its generator streams — delivered WITHOUT blocking the host.  Indexing a GPU tensor with a pageable CPU index makes
    torch upload it with a blocking copy, which on ROCm first drains the stream: one host / GPU synchronisation per
    training step, the host then never runs ahead of the GPU and every step starts with an idle GPU (0.3-0.5 ms; found in
    round 3 with scripts/host_phases.py).  Here the permutations are drawn straight into the rows of one pinned buffer
    (same generator, same order, same values) and uploaded with one asynchronous copy; a ring of buffers with one event
    each keeps a buffer from being redrawn before its upload has run."""
    if device is None or torch.device(device).type != "cuda":
        return torch.stack([torch.randperm(k) for _ in range(count)], dim=0)
    ring = _PINNED_RING.setdefault((k, count, str(torch.device(device))), {"slots": [], "next": 0})
    if len(ring["slots"]) < _PINNED_DEPTH:
        ring["slots"].append([torch.empty(count, k, dtype=torch.int64).pin_memory(), None])
        slot = ring["slots"][-1]
    else:
        slot = ring["slots"][ring["next"]]
        ring["next"] = (ring["next"] + 1) % _PINNED_DEPTH
        # (query first: on ROCm Event.synchronize() of an event that completed long ago still cost the host 0.34 ms per step,
        # scripts/host_profile.py, round 6; it only has to wait when the host is more than _PINNED_DEPTH draws ahead of the GPU)
        if not slot[1].query():
            slot[1].synchronize()
    for s in range(count):
        torch.randperm(k, out=slot[0][s])
    dev = slot[0].to(device, non_blocking=True)
    slot[1] = torch.cuda.Event()
    slot[1].record(torch.cuda.current_stream(device))  # the stream the upload was enqueued on
    return dev


def draw_subnetwork_permutations(batch: int, num_subnetworks: int, input_repetition_probability: float = 0.0,
                                 batch_repetitions: int = 1, device=None) -> torch.Tensor:
    """[S, batch*reps] int64 gather indices: a main permutation of the batch, of which the first
    (1 - irp) share is re-shuffled independently per subnetwork (utils.py:27-36).  One upload and one gather for all S
    rows (round 4: S uploads, S gathers, S concatenations and a stack were ~8 five-microsecond launches per step)."""
    main = torch.randperm(batch, device=device)
    if batch_repetitions != 1:
        main = main.repeat(batch_repetitions)
    k = int(main.shape[0] * (1.0 - input_repetition_probability))
    rows = main[:k][_cpu_randperms_on_device(k, num_subnetworks, main.device)]  # [S, k]
    if k == main.shape[0]:
        return rows
    return torch.cat((rows, main[k:].unsqueeze(0).expand(num_subnetworks, -1)), dim=1)


def gather_subnetworks(t: Optional[torch.Tensor], perms: torch.Tensor) -> Optional[torch.Tensor]:
    """[B,C,H,W] -> [B',S,C,H,W] with t[perms[s]] on subnetwork s."""
    if t is None:
        return None
    return torch.stack([torch.index_select(t, 0, perms[s]) for s in range(perms.shape[0])], dim=1)


def apply_input_transform(image: torch.Tensor, label: torch.Tensor, mask: Optional[torch.Tensor], num_subnetworks: int,
                          input_repetition_probability: float = 0.0, batch_repetitions: int = 1):
    perms = draw_subnetwork_permutations(image.shape[0], num_subnetworks, input_repetition_probability,
                                         batch_repetitions, device=image.device)
    return gather_subnetworks(image, perms), gather_subnetworks(label, perms), gather_subnetworks(mask, perms)


def repeat_subnetworks(x: torch.Tensor, num_subnetworks: int) -> torch.Tensor:
    """[B,C,H,W] -> [B,S,C,H,W]."""
    return x[:, None, :, :, :].repeat(1, num_subnetworks, 1, 1, 1)


def flatten_subnetwork_dimension(x: torch.Tensor) -> torch.Tensor:
    """[B,S,C,H,W] -> [B*S,C,H,W]."""
    b, s, c, h, w = x.shape
    return x.reshape(b * s, c, h, w)


def compute_uncertainties(criterion, y_preds: torch.Tensor, log_params: torch.Tensor):
    """(mean, aleatoric_variance, epistemic_variance), each [B,C,H,W]  (utils.py:76-101), computed by
    the fused reduction kernel `mimo_uncertainties`.  Host tensors (the reference hands this function
    CPU tensors, ensemble.py:104-113) are staged through the GPU and returned on the host."""
    from ..engine import uncertainties
    y_preds, log_params = y_preds.detach().float(), log_params.detach().float()
    if y_preds.is_cuda:
        return uncertainties(y_preds, log_params, criterion.name)
    if not torch.cuda.is_available():
        from .._lib import MimoHipError
        raise MimoHipError("compute_uncertainties runs on an AMD GPU (mimo_uncertainties); no GPU is visible")
    return tuple(t.cpu() for t in uncertainties(y_preds.cuda(), log_params.cuda(), criterion.name))
