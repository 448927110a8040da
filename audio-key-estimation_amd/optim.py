"""Fused Adam for ``PitchClassNet`` (reference optimizer: models.py:1017-1027).

One ``ake_adam_step_f32`` launch over the module's flat parameter buffer replaces torch.optim.Adam's per-tensor passes over
~100 small tensors; the kernel's arithmetic follows torch's ``_single_tensor_adam`` (tests/test_gpu_training.py compares
them step by step).  Subclasses ``torch.optim.Optimizer`` so that ``param_groups``, ``zero_grad`` and the
``ExponentialLR`` scheduler the reference attaches keep working.

Deviation from ``torch.optim.Adam`` (documented): every trainable element is updated on every step
with ONE global step count.  torch skips a parameter whose ``.grad`` is None and counts steps per parameter.  On this path a
gradient is never None: the genre loss is a masked mean on the device (no host sync), so the genre head of a batch without
genre labels has an exact-zero gradient, as it has in the reference's pinned environment (torch 1.8 / Lightning 1.6.4:
``zero_grad()`` zero-fills, requirements.txt:235,244) from the first labelled batch on.  The difference is confined to the
steps BEFORE the first genre-labelled batch (torch: genre head untouched, its bias correction starts later) and to
``--reg > 0`` there (decay applied here).
"""
from __future__ import annotations

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, net, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(list(net.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.net = net
        self.step_count = 0
        self.grad_scale = 1.0          # e.g. 1/world_size after a summed all-reduce of the flat gradient buffer
        self.exp_avg = self.exp_avg_sq = self._trainable = None
        self._trainable_names = None

    def _buffers(self):
        flat, grad = self.net.flat_parameters()
        if not self.net._attached:
            raise _lib.AkeError("FusedAdam needs float32 parameters on the HIP device (they live in one flat buffer); "
                                "use torch.optim.Adam for other dtypes")
        if self.exp_avg is None or self.exp_avg.data_ptr() == 0 or self.exp_avg.device != flat.device or self.exp_avg.numel() != flat.numel():
            self.exp_avg, self.exp_avg_sq = torch.zeros_like(flat), torch.zeros_like(flat)
            self._trainable_names = None
        names = frozenset(n for n, p in self.net.named_parameters() if p.requires_grad)
        if names != self._trainable_names:                     # requires_grad may change between steps (freezing a head)
            mask = torch.zeros(flat.numel(), dtype=torch.uint8)
            for name, off, cnt in self.net._layout():
                if name in names:
                    mask[off:off + cnt] = 1
            self._trainable, self._trainable_names = mask.to(flat.device), names
        return flat, grad

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        flat, grad = self._buffers()
        net = self.net
        if not net._grads_in_place():
            # gradients arrived some other way (foreign .grad tensors): gather them into the flat layout
            offs = net._grad_offsets()
            grad = torch.zeros_like(flat)
            for n, p in net.named_parameters():
                if p.grad is not None:
                    grad[offs[n]:offs[n] + p.numel()].copy_(p.grad.reshape(-1))
        else:
            grad = net._flat_grad
        g = self.param_groups[0]
        self.step_count += 1
        with torch.cuda.device(flat.device):
            _lib.check(_lib.lib().ake_adam_step_f32(flat.data_ptr(), grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                                    self._trainable.data_ptr(), flat.numel(), g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                                                    g["weight_decay"], self.step_count, self.grad_scale,
                                                    torch.cuda.current_stream().cuda_stream), "ake_adam_step_f32")
        net.mark_parameters_changed()
        return loss

    def zero_grad(self, set_to_none: bool = False):
        """One memset of the flat gradient buffer (the p.grad views stay in place)."""
        if self.net._attached and self.net._flat_grad is not None and self.net._grads_in_place():
            self.net._flat_grad.zero_()
        else:
            super().zero_grad(set_to_none=set_to_none)

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq,
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self._buffers()
        self.step_count = int(sd["step"])
        if sd["exp_avg"] is not None:
            self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            g.update(saved)
