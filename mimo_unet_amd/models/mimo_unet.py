"""`MimoUnetModel`: the LightningModule surface of the reference
(``mimo/models/mimo_unet.py:15-314``) on top of the HIP engine.

Same 19 constructor arguments, same attributes (`num_subnetworks`, `filter_base_count`,
`in_channels`, `out_channels`, `loss_fn`, `loss_buffer`, `model`), same
`forward / training_step / validation_step / configure_optimizers /
add_model_specific_args`, same logged names and step-output dict keys, so the
reference's `scripts/train/*` and `scripts/test/*` call it unchanged.

What differs underneath: `training_step` does not materialise the S shuffled copies of
the batch — the permutation indices go to the engine, which gathers while loading
(`mimo_forward` perm argument) and evaluates the NLL and its gradient inside the head
kernels (`mimo_loss_forward`, `mimo_backward` dloss argument)."""
from __future__ import annotations

from argparse import ArgumentParser
from typing import Any, Dict, Literal, Optional, Tuple

import torch

from ..engine import training_epilogue, validation_epilogue
from ..lightning_compat import LightningModule
from ..losses import UncertaintyLoss
from ..metrics import compute_regression_metrics
from ..optim import FlatAdam
from ..utils import count_trainable_parameters
from .mimo_components.loss_buffer import LossBuffer
from .mimo_components.model import MimoUNet
from .utils import (compute_uncertainties, draw_subnetwork_permutations, flatten_subnetwork_dimension,
                    gather_subnetworks)


class MimoUnetModel(LightningModule):
    def __init__(
            self,
            in_channels: int,
            out_channels: int,
            num_subnetworks: int,
            filter_base_count: int,
            center_dropout_rate: float,
            final_dropout_rate: float,
            encoder_dropout_rate: float,
            core_dropout_rate: float,
            decoder_dropout_rate: float,
            loss: str,
            weight_decay: float,
            learning_rate: float,
            seed: int,
            loss_buffer_size: int,
            loss_buffer_temperature: float,
            input_repetition_probability: float = 0.0,
            batch_repetitions: int = 1,
            scheduler_step_size: int = 20,
            scheduler_gamma: float = 0.5,
    ):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.num_subnetworks = num_subnetworks
        self.filter_base_count = filter_base_count
        self.center_dropout_rate = center_dropout_rate
        self.final_dropout_rate = final_dropout_rate
        self.encoder_dropout_rate = encoder_dropout_rate
        self.core_dropout_rate = core_dropout_rate
        self.decoder_dropout_rate = decoder_dropout_rate

        self.loss_fn = UncertaintyLoss.from_name(loss)
        self.loss_name = loss
        self.weight_decay = weight_decay
        self.learning_rate = learning_rate
        self.seed = seed
        self.loss_buffer_size = loss_buffer_size
        self.loss_buffer_temperature = loss_buffer_temperature
        self.input_repetition_probability = input_repetition_probability
        self.batch_repetitions = batch_repetitions
        self.scheduler_step_size = scheduler_step_size
        self.scheduler_gamma = scheduler_gamma

        self.model = MimoUNet(
            in_channels=in_channels, out_channels=out_channels, num_subnetworks=num_subnetworks,
            filter_base_count=filter_base_count, center_dropout_rate=center_dropout_rate,
            final_dropout_rate=final_dropout_rate, encoder_dropout_rate=encoder_dropout_rate,
            core_dropout_rate=core_dropout_rate, decoder_dropout_rate=decoder_dropout_rate,
            bilinear=True, use_pooling_indices=False, loss=loss)
        self.loss_buffer = LossBuffer(buffer_size=loss_buffer_size, temperature=loss_buffer_temperature,
                                      subnetworks=num_subnetworks)
        self.save_hyperparameters()
        self.save_hyperparameters({"loss": loss, "trainable_params": count_trainable_parameters(self.model)})
        self.use_fused_optimizer = True

    # The reference wraps the net in torch.compile (mimo_unet.py:89-91).  The engine is already a
    # fixed kernel schedule; nothing to trace.  Checkpoints written by a compiled reference model
    # carry an `_orig_mod.` infix, accepted on load.
    def compile(self, *args, **kwargs):
        return self

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        cleaned = {k.replace("model._orig_mod.", "model."): v for k, v in state_dict.items()}
        return super().load_state_dict(cleaned, strict=strict, **kw)

    def forward(self, x: torch.Tensor):
        """x [B,S,C_in,H,W] -> (p1, p2), each [B,S,C_out/2,H,W]."""
        _, S, C_in, _, _ = x.shape
        assert S == self.num_subnetworks, "subnetwork dimension must match num_subnetworks"
        assert C_in == self.in_channels, "channel dimension must match in_channels"
        out = self.model(x)
        half = self.out_channels // 2
        return out[:, :, :half, ...], out[:, :, half:, ...]

    # ------------------------------------------------------------------ training -------------
    def training_step(self, batch: Dict[str, torch.Tensor], batch_idx: int) -> Dict[str, torch.Tensor]:
        image, label = batch["image"], batch["label"]
        mask = batch["mask"] if "mask" in batch else None
        perms = draw_subnetwork_permutations(image.shape[0], self.num_subnetworks, self.input_repetition_probability,
                                             self.batch_repetitions, device=image.device)
        return self.training_step_with_perms(image, label, mask, perms)

    def training_step_with_perms(self, image, label, mask, perms) -> Dict[str, torch.Tensor]:
        out, loss = self.model.forward_with_loss(image, label, mask, perms)
        half = self.out_channels // 2
        p1, p2 = out[:, :, :half, ...], out[:, :, half:, ...]
        fused = self.loss_buffer.step(loss)  # weights read BEFORE the add (mimo_unet.py:243-245), one kernel
        if fused is not None:
            loss_mean, weights, train_loss = fused
        else:
            weights = self.loss_buffer.get_weights().to(loss.device)
            self.loss_buffer.add(loss.detach())
            loss_mean = (loss * weights).mean()  # the differentiable scalar, outside the no_grad section below
            train_loss = None
        with torch.no_grad():
            mask_t = gather_subnetworks(mask, perms)
            self._log_train_loss_and_weights(loss.detach(), weights, train_loss)
            if out.is_cuda and self.loss_name in ("laplace_nll", "gaussian_nll"):
                # label gather, mode / std, error map and the regression metrics in one pass
                # (engine.training_epilogue; the reference runs ~20 full-tensor torch ops here)
                label_t, y_pred, aleatoric_std, err_map, sc = training_epilogue(out, label, perms, self.loss_name)
                for i, name in ((3, "r2"), (0, "mae"), (1, "mse"), (2, "rmse")):
                    self._log(f"metric_train/{name}", sc[i], on_step=True, on_epoch=True)
                return {
                    "loss": loss_mean,
                    "label": flatten_subnetwork_dimension(label_t),
                    "preds": flatten_subnetwork_dimension(y_pred),
                    "aleatoric_std_map": flatten_subnetwork_dimension(aleatoric_std),
                    "err_map": flatten_subnetwork_dimension(err_map),
                    "mask": flatten_subnetwork_dimension(mask_t) if mask_t is not None else None,
                }
            label_t = gather_subnetworks(label, perms)
            y_pred = self.loss_fn.mode(p1, p2).detach()
            aleatoric_std = self.loss_fn.std(p1, p2).detach()
            self._log_metrics(y_pred=y_pred, y_true=label_t, stage="train")
        return {
            "loss": loss_mean,
            "label": flatten_subnetwork_dimension(label_t),
            "preds": flatten_subnetwork_dimension(y_pred),
            "aleatoric_std_map": flatten_subnetwork_dimension(aleatoric_std),
            "err_map": flatten_subnetwork_dimension(y_pred - label_t),
            "mask": flatten_subnetwork_dimension(mask_t) if mask_t is not None else None,
        }

    def validation_step(self, batch: Dict[str, torch.Tensor], batch_idx: int) -> Dict[str, torch.Tensor]:
        image, label = batch["image"], batch["label"]
        mask = batch["mask"] if "mask" in batch else None
        S = self.num_subnetworks
        with torch.no_grad():
            out, val_loss = self._val_forward(image, label, mask)
            # uncertainties, combined NLL on the ensemble mean, error map and the regression metrics in one pass
            # (engine.validation_epilogue; the reference runs ~15 full-tensor torch ops here)
            y_mean = label
            y_pred_mean, aleatoric_std, epistemic_std, err_map, sc = validation_epilogue(
                out, label, mask, self.loss_name, self.loss_fn.eps_min, self.loss_fn.eps_max)
            self._log_val_loss(val_loss, sc[0])
            for i, name in ((4, "r2"), (1, "mae"), (2, "mse"), (3, "rmse")):
                self._log(f"metric_val/{name}", sc[i], on_step=False, on_epoch=True)
            self._log("metric_val/aleatoric_std_mean", sc[5])
            self._log("metric_val/epistemic_std_mean", sc[6])
        return {
            "loss": val_loss.mean(),
            "label": y_mean,
            "preds": y_pred_mean,
            "aleatoric_std_map": aleatoric_std,
            "epistemic_std_map": epistemic_std,
            "err_map": err_map,
            "mask": mask,
        }

    def _val_forward(self, image, label, mask):
        """Every subnetwork sees the same image/label (repeat_subnetworks, mimo_unet.py:150-152):
        identity permutations let the engine broadcast instead of copying."""
        n = image.shape[0]
        ident = torch.arange(n, device=image.device, dtype=torch.int64)[None].repeat(self.num_subnetworks, 1)
        return self.model.forward_with_loss(image, label, mask, ident)

    # Lightning epoch-end hooks: one device synchronisation per epoch turns a diverged / out-of-range run into an
    # error that says what happened (MimoUNet.check_numerics) instead of NaN losses scrolling by
    def on_train_epoch_end(self) -> None:
        self.model.check_numerics()

    def on_validation_epoch_end(self) -> None:
        self.model.check_numerics()

    # data parallel: BatchNorm running statistics are per rank; checkpoints and validation use rank 0's (ddp.py)
    def on_save_checkpoint(self, checkpoint) -> None:
        from ..ddp import broadcast_buffers
        broadcast_buffers(self.model)

    def on_validation_epoch_start(self) -> None:
        from ..ddp import broadcast_buffers
        broadcast_buffers(self.model)

    def configure_optimizers(self) -> Dict[str, Any]:
        if self.use_fused_optimizer:
            optimizer = FlatAdam(self.model, lr=self.learning_rate, weight_decay=self.weight_decay)
        else:
            optimizer = torch.optim.Adam(self.parameters(), lr=self.learning_rate, weight_decay=self.weight_decay)
        scheduler = torch.optim.lr_scheduler.StepLR(optimizer, step_size=self.scheduler_step_size,
                                                    gamma=self.scheduler_gamma)
        return dict(optimizer=optimizer, lr_scheduler=scheduler, monitor="val_loss")

    # ------------------------------------------------------------------ logging ----------------
    def _attached_trainer(self):
        """The trainer, or None in hand-written loops / benchmarks (Lightning's `trainer` property raises
        RuntimeError while the module is not attached; the stand-in base class holds None)."""
        try:
            return self.trainer
        except RuntimeError:
            return None

    def _batch_size(self) -> Optional[int]:
        # the reference reads self.trainer.datamodule.batch_size (mimo_unet.py:250)
        return getattr(getattr(self._attached_trainer(), "datamodule", None), "batch_size", None)

    def _log(self, name, value, **kw):
        self.log(name, value, batch_size=self._batch_size(), **kw)

    def _log_train_loss_and_weights(self, loss: torch.Tensor, weights: torch.Tensor, loss_mean: Optional[torch.Tensor] = None) -> None:
        self._log("train_loss", loss.mean() if loss_mean is None else loss_mean)
        for i in range(loss.shape[0]):
            self._log(f"train_loss_{i}", loss[i])
            self._log(f"train_weight_{i}", weights[i])

    def _log_metrics(self, y_pred: torch.Tensor, y_true: torch.Tensor, stage: Literal["train", "val"] = "train") -> None:
        for name, value in compute_regression_metrics(y_pred.flatten(), y_true.flatten()).items():
            self._log(f"metric_{stage}/{name}", value, on_step=(stage == "train"), on_epoch=True)

    def _log_val_loss(self, val_loss: torch.Tensor, val_loss_combined: torch.Tensor) -> None:
        self._log("val_loss", val_loss.mean())
        for i in range(val_loss.shape[0]):
            self._log(f"val_loss_{i}", val_loss[i])
        self._log("val_loss_combined", val_loss_combined)

    def _log_uncertainties(self, aleatoric_std: torch.Tensor, epistemic_std: torch.Tensor) -> None:
        self._log("metric_val/aleatoric_std_mean", aleatoric_std.clip(0, 5).mean())
        self._log("metric_val/epistemic_std_mean", epistemic_std.clip(0, 5).mean())

    @staticmethod
    def add_model_specific_args(parent_parser: ArgumentParser) -> ArgumentParser:
        parser = parent_parser.add_argument_group(title="MIMO UNet Model")
        for name, typ, default in (
                ("num_subnetworks", int, 3), ("filter_base_count", int, 32), ("center_dropout_rate", float, 0.0),
                ("final_dropout_rate", float, 0.0), ("encoder_dropout_rate", float, 0.0),
                ("core_dropout_rate", float, 0.0), ("decoder_dropout_rate", float, 0.0),
                ("input_repetition_probability", float, 0.0), ("batch_repetitions", int, 1),
                ("loss", str, "laplace_nll"), ("learning_rate", float, 1e-3), ("weight_decay", float, 0.0),
                ("loss_buffer_size", int, 10), ("loss_buffer_temperature", float, 1.0),
                ("scheduler_step_size", int, 20), ("scheduler_gamma", float, 0.5)):
            parser.add_argument(f"--{name}", type=typ, default=default)
        return parent_parser
