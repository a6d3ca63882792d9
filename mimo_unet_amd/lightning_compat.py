"""`LightningModule` base: the real one when `lightning` is installed, otherwise a minimal
stand-in with the methods the MIMO modules and the reference's scripts rely on
(`save_hyperparameters`, `hparams`, `log`, `device`, `load_from_checkpoint`)."""
from __future__ import annotations

import inspect
from typing import Any, Dict

import torch
from torch import nn

try:  # pragma: no cover - not installed in the build image
    import lightning.pytorch as pl

    LightningModule = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # ModuleNotFoundError
    HAVE_LIGHTNING = False

    class AttributeDict(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

        def __setattr__(self, k, v):
            self[k] = v

    class LightningModule(nn.Module):
        """Just enough of lightning.pytorch.LightningModule for training/eval loops written by hand."""

        def __init__(self) -> None:
            super().__init__()
            self._hparams = AttributeDict()
            self.logged: Dict[str, Any] = {}
            self.trainer = None

        @property
        def hparams(self) -> AttributeDict:
            return self._hparams

        @property
        def device(self) -> torch.device:
            for p in self.parameters():
                return p.device
            return torch.device("cpu")

        def save_hyperparameters(self, *args) -> None:
            if args and isinstance(args[0], dict):
                self._hparams.update(args[0])
                return
            frame = inspect.currentframe().f_back
            init = getattr(type(self), "__init__")
            names = [n for n in inspect.signature(init).parameters if n != "self"]
            loc = frame.f_locals
            self._hparams.update({n: loc[n] for n in names if n in loc})

        def log(self, name: str, value, **kwargs) -> None:
            self.logged[name] = value.detach() if isinstance(value, torch.Tensor) else value

        def log_dict(self, d, **kwargs) -> None:
            for k, v in d.items():
                self.log(k, v)

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict: bool = True, **overrides):
            ckpt = torch.load(checkpoint_path, map_location=map_location or "cpu", weights_only=False)
            hp = dict(ckpt.get("hyper_parameters", {}))
            hp.update(overrides)
            accepted = inspect.signature(cls.__init__).parameters
            model = cls(**{k: v for k, v in hp.items() if k in accepted})
            model.load_state_dict(ckpt["state_dict"], strict=strict)
            return model

        def checkpoint_dict(self) -> Dict[str, Any]:
            """What `Trainer.save_checkpoint` would persist for `load_from_checkpoint`."""
            return {"state_dict": self.state_dict(), "hyper_parameters": dict(self.hparams)}
