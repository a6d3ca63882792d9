"""Small helpers with the reference's names (mimo/utils.py:4-14)."""
from pathlib import Path


def dir_path(string) -> Path:
    path = Path(string)
    if not path.is_dir():
        raise NotADirectoryError(string)
    return path


def count_trainable_parameters(model) -> int:
    return sum(p.numel() for p in model.parameters() if p.requires_grad)
