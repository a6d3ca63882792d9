"""MIMO batch helpers with the reference's names (``mimo/models/utils.py:5-101``).

On the hot path the gather of `apply_input_transform` is fused into the first kernel of
libmimo_hip.so (``perm`` argument of ``mimo_forward``); `draw_subnetwork_permutations`
produces exactly the index tensor the reference draws.  The functions below keep the
reference's call signatures for callers that want materialised tensors."""
from typing import Optional

import torch


def draw_subnetwork_permutations(batch: int, num_subnetworks: int, input_repetition_probability: float = 0.0,
                                 batch_repetitions: int = 1, device=None) -> torch.Tensor:
    """[S, batch*reps] int64 gather indices: a main permutation of the batch, of which the first
    (1 - irp) share is re-shuffled independently per subnetwork (utils.py:27-36)."""
    main = torch.randperm(batch, device=device).repeat(batch_repetitions)
    k = int(main.shape[0] * (1.0 - input_repetition_probability))
    rows = [torch.cat((main[:k][torch.randperm(k)], main[k:]), dim=0) for _ in range(num_subnetworks)]
    return torch.stack(rows, dim=0)


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
