"""`EvidentialUnetModel` with the reference's constructor, step outputs and logged names
(``mimo/models/evidential_unet.py:13-209``): a single-subnetwork MIMO U-Net backbone with four output
channels turned into Normal-Inverse-Gamma parameters (softplus heads) and trained with the evidential
loss.  The backbone is the HIP engine (`MimoUNet`, S = 1, through its generic autograd bridge); the
softplus heads and the loss are one fused HIP kernel over the [B,4,H,W] logits (one more for their gradient)."""
from __future__ import annotations

from argparse import ArgumentParser
from typing import Any, Dict, Literal

import torch
import torch.nn.functional as F

from ..engine import evidential_head_loss
from ..lightning_compat import LightningModule
from ..losses import EvidentialLoss
from ..metrics import compute_regression_metrics
from ..optim import FlatAdam
from ..utils import count_trainable_parameters
from .mimo_components.model import MimoUNet


class EvidentialUnetModel(LightningModule):
    def __init__(self, in_channels: int, out_channels: int, filter_base_count: int, center_dropout_rate: float,
                 final_dropout_rate: float, encoder_dropout_rate: float, core_dropout_rate: float,
                 decoder_dropout_rate: float, weight_decay: float, learning_rate: float, seed: int,
                 scheduler_step_size: int = 20, scheduler_gamma: float = 0.5):
        super().__init__()
        self.in_channels, self.out_channels, self.filter_base_count = in_channels, out_channels, filter_base_count
        self.center_dropout_rate, self.final_dropout_rate = center_dropout_rate, final_dropout_rate
        self.encoder_dropout_rate, self.core_dropout_rate = encoder_dropout_rate, core_dropout_rate
        self.decoder_dropout_rate = decoder_dropout_rate
        self.loss_fn = EvidentialLoss(coeff=1.0)
        self.weight_decay, self.learning_rate, self.seed = weight_decay, learning_rate, seed
        self.scheduler_step_size, self.scheduler_gamma = scheduler_step_size, scheduler_gamma
        self.use_fused_optimizer = True  # FlatAdam (one launch) instead of torch.optim.Adam; same update rule
        if out_channels != EvidentialLoss.num_distribution_params:
            raise ValueError("the evidential head needs out_channels == 4 (gamma, v, alpha, beta)")
        self.model = MimoUNet(in_channels=in_channels, out_channels=out_channels, num_subnetworks=1,
                              filter_base_count=filter_base_count, center_dropout_rate=center_dropout_rate,
                              final_dropout_rate=final_dropout_rate, encoder_dropout_rate=encoder_dropout_rate,
                              core_dropout_rate=core_dropout_rate, decoder_dropout_rate=decoder_dropout_rate,
                              bilinear=True, use_pooling_indices=False)
        self.save_hyperparameters()
        self.save_hyperparameters({"loss": "evidential", "trainable_params": count_trainable_parameters(self.model)})

    def compile(self):
        """The reference wraps the model in torch.compile here; the HIP engine is already compiled."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, C_in, H, W] -> [B, 4, H, W] = (gamma, v, alpha, beta)."""
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError("channel dimension must match in_channels")
        return evidential_head_loss(self._logits(x))[0]  # the four heads in one kernel (and one in the backward)

    def _logits(self, x: torch.Tensor) -> torch.Tensor:
        return self.model(x.unsqueeze(1)).squeeze(1)

    def _forward_with_loss(self, image: torch.Tensor, label: torch.Tensor, mask):
        """(NIG parameters [B,4,H,W], per-pixel loss [B,H,W]): heads + EvidentialLoss.forward fused into one pass over
        the logits (mimo_evidential_forward); other loss settings fall back to the loss class on the parameters."""
        b, _, h, w = image.shape
        # (a [B,1,H,W] mask would broadcast the reference's [B,H,W] loss map to [B,B,H,W]: left to the loss class)
        if tuple(label.shape) == (b, 1, h, w) and (mask is None or tuple(mask.shape) == (b, h, w)):
            return evidential_head_loss(self._logits(image), label, mask)
        out = self(image)
        return out, self.loss_fn(out, label, mask=mask)

    def training_step(self, batch: Dict[str, torch.Tensor], batch_idx: int) -> Dict[str, torch.Tensor]:
        image, label = batch["image"], batch["label"]
        mask = batch["mask"] if "mask" in batch else None
        out, loss = self._forward_with_loss(image, label, mask)
        y_pred = self.loss_fn.mode(out).unsqueeze(dim=1)
        aleatoric_std = self.loss_fn.aleatoric_var(out).unsqueeze(dim=1) ** 0.5
        self._log_metrics(y_pred=y_pred, y_true=label, stage="train")
        return {"loss": loss.mean(), "label": label, "preds": y_pred, "aleatoric_std_map": aleatoric_std,
                "err_map": y_pred - label, "mask": mask}

    def validation_step(self, batch: Dict[str, torch.Tensor], batch_idx: int) -> Dict[str, torch.Tensor]:
        image, label = batch["image"], batch["label"]
        mask = batch["mask"] if "mask" in batch else None
        with torch.no_grad():
            out, loss = self._forward_with_loss(image, label, mask)
            y_pred = self.loss_fn.mode(out).unsqueeze(dim=1)
            aleatoric_std = self.loss_fn.aleatoric_var(out).unsqueeze(dim=1) ** 0.5
            epistemic_std = self.loss_fn.epistemic_var(out).unsqueeze(dim=1) ** 0.5
            self._log("val_loss", loss.mean())
            self._log_metrics(y_pred=y_pred, y_true=label, stage="val")
            self._log("metric_val/aleatoric_std_mean", aleatoric_std.clip(0, 5).mean())
            self._log("metric_val/epistemic_std_mean", epistemic_std.clip(0, 5).mean())
        return {"loss": loss.mean(), "label": label, "preds": y_pred, "aleatoric_std_map": aleatoric_std,
                "epistemic_std_map": epistemic_std, "err_map": y_pred - label, "mask": mask}

    def on_train_epoch_end(self) -> None:
        self.model.check_numerics()  # see MimoUnetModel.on_train_epoch_end

    def on_validation_epoch_end(self) -> None:
        self.model.check_numerics()

    def on_save_checkpoint(self, checkpoint) -> None:  # see MimoUnetModel: rank 0's BatchNorm buffers
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

    def _log(self, name, value, **kw):
        self.log(name, value, **kw)

    def _log_metrics(self, y_pred: torch.Tensor, y_true: torch.Tensor, stage: Literal["train", "val"] = "train") -> None:
        for name, value in compute_regression_metrics(y_pred.flatten(), y_true.flatten()).items():
            self._log(f"metric_{stage}/{name}", value, on_step=(stage == "train"), on_epoch=True)

    @staticmethod
    def add_model_specific_args(parent_parser: ArgumentParser) -> ArgumentParser:
        parser = parent_parser.add_argument_group(title="MIMO UNet Model")
        for name, typ, default in (("filter_base_count", int, 32), ("center_dropout_rate", float, 0.0),
                                   ("final_dropout_rate", float, 0.0), ("encoder_dropout_rate", float, 0.0),
                                   ("core_dropout_rate", float, 0.0), ("decoder_dropout_rate", float, 0.0),
                                   ("learning_rate", float, 1e-3), ("weight_decay", float, 0.0),
                                   ("scheduler_step_size", int, 20), ("scheduler_gamma", float, 0.5)):
            parser.add_argument(f"--{name}", type=typ, default=default)
        return parent_parser
