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
    """``fit`` / ``validate`` loops with the Lightning 1.6 hook order (train_model.py:112-124, eval.py:118-129).

    ``fit`` mirrors what ``pl.Trainer(max_epochs, accumulate_grad_batches).fit(model)`` does to the reference module:
    ``configure_optimizers`` once; per batch ``training_step`` -> ``(loss / accumulate_grad_batches).backward()``; an
    optimizer step every ``accumulate_grad_batches`` batches and on the last batch of the epoch; the LR scheduler once
    per epoch; then the validation loop and ``validation_epoch_end``.  Under torch.distributed (one process per GPU)
    the gradients are all-reduced right before each optimizer step (``distributed.all_reduce_gradients``).
    Callbacks (EarlyStopping / ModelCheckpoint) are not re-implemented: the module saves its own best checkpoint
    (models.py:991-993)."""

    def __init__(self, max_epochs=1, accumulate_grad_batches=1, logger=None, **_ignored):
        self.max_epochs = max_epochs
        self.accumulate_grad_batches = accumulate_grad_batches
        self.logger = logger or NullLogger()
        self.train_losses = []          # per training batch (floats), filled at the end of each epoch
        self.val_results = []

    def fit(self, model, train_dataloaders=None, val_dataloaders=None, max_steps=None):
        from . import distributed as D
        self._attach(model)
        device = next(model.parameters()).device
        optimizers, schedulers = model.configure_optimizers()
        opt = optimizers[0]
        sched = schedulers[0] if schedulers else None
        acc = max(1, int(self.accumulate_grad_batches))
        D.broadcast_parameters(model)
        model.global_step = 0
        for epoch in range(self.max_epochs):
            model.train()
            loader = train_dataloaders if train_dataloaders is not None else model.train_dataloader()
            n = len(loader)
            opt.zero_grad()
            losses = []
            for i, batch in enumerate(loader):
                out = model.training_step(_to_device(batch, device), i)
                (out["loss"] / acc).backward()
                losses.append(out["loss"].detach())
                if (i + 1) % acc == 0 or i + 1 == n:
                    scale = D.all_reduce_gradients(model)
                    if hasattr(opt, "grad_scale"):
                        opt.grad_scale = scale
                    elif scale != 1.0:
                        for p in model.parameters():
                            if p.grad is not None:
                                p.grad.mul_(scale)
                    opt.step()
                    opt.zero_grad()
                    model.global_step += 1
                    if max_steps is not None and model.global_step >= max_steps:
                        break
            self.train_losses += [float(v) for v in torch.stack(losses).cpu()]
            if sched is not None:
                sched.step()
            if val_dataloaders is not None or (getattr(model, "data", None) or {}).get("val") is not None:
                self.val_results.append(self.validate(model, val_dataloaders)[0])
            if max_steps is not None and model.global_step >= max_steps:
                break
        return self

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
