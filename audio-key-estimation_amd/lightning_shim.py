"""Minimal stand-in for the slice of pytorch_lightning 1.6 the reference scripts touch.

pytorch_lightning is not installed on a fresh ROCm box, and the reference only uses
``LightningModule`` hooks plus ``Trainer(...).validate/fit`` (train_model.py:110-124,
eval.py:104-129).  If the real package is importable we use it; otherwise these ~80 lines
provide the same hook protocol (``validation_step`` -> ``validation_epoch_end`` -> ``self.log``)
so the drop-in ``PitchClassNet`` can be driven the same way.
"""
from __future__ import annotations

import torch
from torch import nn

try:                                   # pragma: no cover - not available in the build image
    import pytorch_lightning as _pl
    LightningModule = _pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:                      # noqa: BLE001
    HAVE_LIGHTNING = False

    class LightningModule(nn.Module):
        def __init__(self):
            super().__init__()
            self.trainer = None
            self.logger = None
            self.global_step = 0
            self.logged = {}

        def log(self, name, value, **kw):
            self.logged[name] = float(value) if not isinstance(value, float) else value


class _NullExperiment:
    def add_scalar(self, *a, **k):
        pass


class NullLogger:
    """Stands in for TensorBoardLogger (train_model.py:113): accepts add_scalar, has a version."""
    version = 0
    experiment = _NullExperiment()


def _to_device(batch, device):
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}


class Trainer:
    """``validate`` / ``test`` loops with the Lightning 1.6 hook order (eval.py:118-129)."""

    def __init__(self, max_epochs=1, accumulate_grad_batches=1, logger=None, **_ignored):
        self.max_epochs = max_epochs
        self.accumulate_grad_batches = accumulate_grad_batches
        self.logger = logger or NullLogger()

    def _attach(self, model):
        model.trainer = self
        if getattr(model, "logger", None) is None:
            model.logger = self.logger

    @torch.no_grad()
    def validate(self, model, dataloaders=None):
        self._attach(model)
        was_training = model.training
        model.eval()
        device = next(model.parameters()).device
        loader = dataloaders if dataloaders is not None else model.val_dataloader()
        outputs = [model.validation_step(_to_device(b, device), i) for i, b in enumerate(loader)]
        result = model.validation_epoch_end(outputs)
        model.train(was_training)
        return [{k: float(v) for k, v in result.items() if k != "log"}]
