"""Drop-in ``PitchClassNet`` whose forward pass runs on hand-written HIP kernels (gfx950).

Same constructor, ``forward(mel, seq_length)`` contract, LightningModule hook names and
``state_dict`` keys/shapes as the reference class (models.py:651-1116), so
``train_model.py`` / ``eval.py`` style code and a reference ``best_model.pt`` load unchanged
(``load_state_dict(strict=True)``, eval.py:115).

The torch modules below are *parameter containers only* (they give the parameters their
reference names and default initialisation); no torch op computes the network.  ``forward``
hands the tensors to ``libake_hip.so`` (``ake_pcnet_forward_f32``); without that library, or on
a CPU tensor, it raises -- there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .lightning_shim import LightningModule
from .metrics import mirex_score as _mirex_score

_VARIANT_FLAGS = ("only_semitones",)
# opt.precision -> ake_pcnet_config::precision (include/ake_hip.h: AKE_PRECISION_MIXED / AKE_PRECISION_F32X3)
PRECISIONS = {"mixed": 0, "f32x3": 1, 0: 0, 1: 1}
PRECISION_DTYPES = {
    0: "f32 accumulation everywhere; pitch / semitone / layer-0 convolutions: f16 activations x f16 weights (per-channel power-of-two scaled) on "
       "MFMA, one product; last layer's pitch-class convolutions and heads: 3-term split-bf16 on MFMA (hi*hi + lo*hi + hi*lo)",
    1: "f32 accumulation everywhere; pitch convolutions: f16 hi + lo operands, three MFMA products (f32-equivalent to 2^-22); semitone / layer-0 "
       "convolutions: exact f32 (v_mfma_f32_16x16x4_f32 / VALU); pitch-class convolutions and heads: 3-term split-bf16 on MFMA (operands to 2^-17)",
}


class EquivariantPitchClassConvolutionSimple(nn.Module):
    """Parameter container for models.py:22-33: ``conv2d`` = Conv2d(cin, cout, (12, kd))."""

    def __init__(self, pitch_classes, in_channels, out_channels, kernel_depth, same_depth_padding=False):
        super().__init__()
        self.conv2d = nn.Conv2d(in_channels, out_channels, (pitch_classes, kernel_depth),
                                padding=(0, kernel_depth // 2 if same_depth_padding else 0))


class _ConvStack(nn.Module):
    """Parameter container for PitchClass2PitchClass / Pitch2Pitch (models.py:190-199, 227-237): ``layer`` Sequential."""

    def __init__(self, blocks):
        super().__init__()
        self.layer = nn.Sequential(*blocks)


class ResBlock(nn.Module):
    """Parameter container for models.py:402-427 (x -> act2(x + b2(conv2(act1(b1(conv1(x))))))); creation order as the reference's."""

    def __init__(self, kernel_size, conv_layers, num_filters):
        super().__init__()
        k = kernel_size
        self.conv1 = nn.Conv2d(num_filters, 2 * num_filters, k, padding=(k // 2, k // 2), padding_mode="circular")
        self.b1 = nn.BatchNorm2d(2 * num_filters)
        self.act1 = nn.LeakyReLU()
        self.conv2 = nn.Conv2d(2 * num_filters, num_filters, k, padding=(k // 2, k // 2), padding_mode="circular")
        self.b2 = nn.BatchNorm2d(num_filters)
        self.act2 = nn.LeakyReLU()


class ResBlockEquivariant(nn.Module):
    """Parameter container for models.py:429-454."""

    def __init__(self, kernel_size, conv_layers, num_filters):
        super().__init__()
        self.conv1 = EquivariantPitchClassConvolutionSimple(12, num_filters, 2 * num_filters, kernel_size, True)
        self.b1 = nn.BatchNorm2d(2 * num_filters)
        self.act1 = nn.LeakyReLU()
        self.conv2 = EquivariantPitchClassConvolutionSimple(12, 2 * num_filters, num_filters, kernel_size, True)
        self.b2 = nn.BatchNorm2d(num_filters)
        self.act2 = nn.LeakyReLU()


class _DenseLayer(nn.Module):
    """Parameter container for models.py:456-471 / :519-534 (--denseblock): norm1 -> LeakyReLU -> 1-wide bottleneck conv -> norm2 -> ReLU
    -> k-wide conv.  ``equivariant``: both convolutions are EquivariantPitchClassConvolutionSimple (12 x 1, 12 x k; with bias), else plain
    Conv2d without bias (1 x 1, k x k zero-padded)."""

    def __init__(self, cin, growth, bn_size, k, equivariant):
        super().__init__()
        self.norm1 = nn.BatchNorm2d(cin)
        self.relu1 = nn.LeakyReLU(inplace=True)
        if equivariant:
            self.conv1 = EquivariantPitchClassConvolutionSimple(12, cin, bn_size * growth, 1)
        else:
            self.conv1 = nn.Conv2d(cin, bn_size * growth, kernel_size=1, stride=1, bias=False)
        self.norm2 = nn.BatchNorm2d(bn_size * growth)
        self.relu2 = nn.ReLU(inplace=True)
        if equivariant:
            self.conv2 = EquivariantPitchClassConvolutionSimple(12, bn_size * growth, growth, k, True)
        else:
            self.conv2 = nn.Conv2d(bn_size * growth, growth, kernel_size=k, stride=1, padding=k // 2, bias=False)


class DenseBlock(nn.ModuleDict):
    """Parameter container for DenseBlock / DenseBlockEquivariant, models.py:584-648: ``denselayer1..n``, layer i reading cin + i*growth channels."""

    def __init__(self, num_layers, cin, bn_size, growth, k, equivariant):
        super().__init__()
        for i in range(num_layers):
            self.add_module("denselayer%d" % (i + 1), _DenseLayer(cin + i * growth, growth, bn_size, k, equivariant))


def _dense_stack(cin, growth, k, n, equivariant):
    """models.py:188-189 / :225-226: bn_size = cin // 2 (1 for a single input channel), multi_path off (:264)."""
    return _ConvStack([DenseBlock(n, cin, cin // 2 if cin > 1 else 1, growth, k, equivariant)])


def _pc2pc(cin, cout, k, n, resblock=False):
    if resblock:                                                      # models.py:181-187
        blocks = [EquivariantPitchClassConvolutionSimple(12, cin, cout, k, True), nn.BatchNorm2d(cout), nn.LeakyReLU()]
        return _ConvStack(blocks + [ResBlockEquivariant(k, n, cout) for _ in range(n)])
    blocks = []
    for i in range(n):
        blocks += [EquivariantPitchClassConvolutionSimple(12, cin if i == 0 else cout, cout, k, True), nn.BatchNorm2d(cout), nn.LeakyReLU()]
    return _ConvStack(blocks)


def _p2p(cin, cout, k, n, resblock=False):
    if resblock:                                                      # models.py:218-224
        blocks = [nn.Conv2d(cin, cout, k, padding=k // 2, padding_mode="circular"), nn.BatchNorm2d(cout), nn.LeakyReLU()]
        return _ConvStack(blocks + [ResBlock(k, n, cout) for _ in range(n)])
    blocks = []
    for i in range(n):
        blocks += [nn.Conv2d(cin if i == 0 else cout, cout, k, padding=k // 2, padding_mode="circular"), nn.BatchNorm2d(cout), nn.LeakyReLU()]
    return _ConvStack(blocks)


class Pitch2PitchClassConv(nn.Module):
    """Parameter container for models.py:108-121 (--p2pc_conv): the octave fold as Conv2d(C, C, (n_oct, 1), dilation (12, 1)) + BN + act."""

    def __init__(self, pitches_in, in_channels):
        super().__init__()
        k = -(-pitches_in // 12)
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=(k, 1), dilation=(12, 1))
        self.bn = nn.BatchNorm2d(in_channels)
        self.act = nn.LeakyReLU()


class PitchClassNetLayer(nn.Module):
    """Parameter container for one layer (models.py:246-350); creation order matches the reference
    so that the same torch seed gives the same initial weights."""

    def __init__(self, layer_num, nf, k, conv_layers, resblock=False, pc2p_mem=False, p2pc_conv=False, pitches=288, stay_sixth=False,
                 denseblock=False):
        super().__init__()
        if denseblock:                                                            # channel algebra of models.py:267-278
            g = nf * conv_layers
            prev_p, prev_pc = 1, 1 + g
            for _ in range(layer_num - 1):
                prev_p += g + prev_pc
                prev_pc += g + prev_p
            if layer_num == 0:
                self.pool_semi = nn.Conv2d(1, 1, 3, stride=(3, 1), padding=(0, 1), padding_mode="circular")
                self.pool_semi_b = nn.BatchNorm2d(1)
                self.pool_semi_a = nn.LeakyReLU()
                self.pc2pc = _dense_stack(1, nf, k, conv_layers, True)
                self.out_pc = 1 + g
                return
            out_p = prev_p + g + prev_pc
            self.up_sixth = nn.ConvTranspose2d(prev_pc, prev_pc, kernel_size=(3, 1), stride=(3, 1))
            self.up_sixth_b = nn.BatchNorm2d(prev_pc)
            self.up_sixth_a = nn.LeakyReLU()
            self.p2p = _dense_stack(prev_pc + prev_p, nf, k, conv_layers, False)
            self.pool_semi = nn.Conv2d(out_p, out_p, (3, 3), stride=(3, 1), padding=(0, 1), padding_mode="circular")
            self.pool_semi_b = nn.BatchNorm2d(out_p)
            self.pool_semi_a = nn.LeakyReLU()
            self.pc2pc = _dense_stack(out_p + prev_pc, nf, k, conv_layers, True)
            self.out_pc = out_p + prev_pc + g                                     # = final_channels of models.py:686-689
            return
        if layer_num == 0:
            self.pool_semi = nn.Conv2d(1, 1, 3, stride=(3, 1), padding=(0, 1), padding_mode="circular")
            self.pool_semi_b = nn.BatchNorm2d(1)
            self.pool_semi_a = nn.LeakyReLU()
            if p2pc_conv:
                self.pool = Pitch2PitchClassConv(pitches // 3, 1)                # models.py:316-317
            self.pc2pc = _pc2pc(1, nf, k, conv_layers, resblock)
            return
        if layer_num == 1:
            prev_p, prev_pc = 1, nf
            out_p = 2 * nf
            out_pc = 2 * out_p
        else:
            prev_p = 2 * nf if layer_num == 2 else 2 * nf * 4 ** (layer_num - 2)
            prev_pc = 2 * prev_p
            out_p, out_pc = 4 * prev_p, 4 * prev_pc
        if not stay_sixth:                                                        # models.py:322-327
            self.up_sixth = nn.ConvTranspose2d(prev_pc, prev_pc, kernel_size=(3, 1), stride=(3, 1))
            self.up_sixth_b = nn.BatchNorm2d(prev_pc)
            self.up_sixth_a = nn.LeakyReLU()
        # --pc2p_mem (models.py:145-166, 333): the up_sixth map is added to the pitch stream, not concatenated -> fewer input channels
        self.p2p = _p2p(prev_p if pc2p_mem else prev_pc + prev_p, out_p, k, conv_layers, resblock)
        if not stay_sixth:                                                        # models.py:336-339
            self.pool_semi = nn.Conv2d(out_p, out_p, (3, 3), stride=(3, 1), padding=(0, 1), padding_mode="circular")
            self.pool_semi_b = nn.BatchNorm2d(out_p)
            self.pool_semi_a = nn.LeakyReLU()
        if p2pc_conv:
            self.pool = Pitch2PitchClassConv(pitches // 3, out_p)                # models.py:340-341
        self.pc2pc = _pc2pc(out_p + prev_pc, out_pc, k, conv_layers, resblock)
        self.out_pc = out_pc


class _TrainStep(torch.autograd.Function):
    """Autograd node for the whole network: forward = ake_pcnet_forward_train_f32, backward = ake_pcnet_backward_f32.

    The parameters are passed as inputs only so that autograd knows the outputs depend on them; no torch op touches the
    activations.  When the parameters live in the module's flat device buffer (the normal case) the backward kernels add
    straight into the flat gradient buffer that every ``p.grad`` is a view of -- nothing is returned to autograd for them.
    """

    @staticmethod
    def forward(ctx, net, x, seq, *params):
        # The backward kernels read the activations this forward leaves in its workspace, so the workspace belongs to THIS node
        # until its backward has run (plain autograd keeps the activations per graph, the reference relies on it: two forwards
        # before one backward, a validation forward in between, a larger batch in between).  It comes from the module's pool of
        # free training workspaces and goes back there after backward; a node that is dropped without backward frees it.
        B, Tn = x.shape[0], x.shape[3]
        ws = net._train_ws_acquire(B, Tn, x.device)
        key, tonic, genre = net._forward_train_raw(x, seq, ws)
        ctx.net, ctx.x, ctx.seq, ctx.key, ctx.ws = net, x, seq, key, ws
        ctx.bn_stats = net._last_bn_stats if net.denseblock else None      # (the dense layers' norm1 blend them a second time in backward)
        ctx.param_meta = [(p.dtype, p.shape) for p in params]
        ctx.set_materialize_grads(True)
        if genre is None:
            return key, tonic
        return key, tonic, genre

    @staticmethod
    def backward(ctx, d_key, d_tonic, d_genre=None):
        net, ws = ctx.net, ctx.ws
        if ws is None:
            raise _lib.AkeError("backward through the same PitchClassNet forward twice: its activations were released after the first "
                                "backward (as autograd does without retain_graph)")
        ctx.ws = None
        try:
            if ctx.bn_stats is not None:
                net._update_recomputed_running_stats(ctx.bn_stats)
                ctx.bn_stats = None
            if net._grads_in_place():
                net._backward_raw(ctx.x, ctx.seq, ctx.key, d_key, d_tonic, d_genre, into=net._flat_grad, ws=ws)
                return (None, None, None) + (None,) * len(ctx.param_meta)
            flat = net._backward_raw(ctx.x, ctx.seq, ctx.key, d_key, d_tonic, d_genre, ws=ws)
        finally:
            net._train_ws_release(ws)
        grads = []
        for (name, _), (dtype, shape) in zip(net.named_parameters(), ctx.param_meta):
            off = net._grad_offsets()[name]
            grads.append(flat[off:off + shape.numel()].view(shape).to(dtype))
        return (None, None, None) + tuple(grads)


class _FusedGeneralStep(torch.autograd.Function):
    """general_step's loss, its gradient w.r.t. the network outputs and the nine metrics in ONE launch (ake_general_step_f32) instead of
    ~110 torch kernels forward and ~50 in autograd's backward (models.py:826-905, 1065-1116).  Returns a float32 tensor of 10 scalars in
    general_step's order; only element 0 (the loss) carries a gradient."""

    @staticmethod
    def forward(ctx, key_out, tonic_out, genre_out, key_labels, tonic_labels, genre_labels, key_signature_id, weights, use_cos):
        dev = key_out.device
        B = key_out.shape[0]

        def onehot(t):
            t = t.to(dev)
            if t.dtype not in (torch.float32, torch.int64):
                t = t.long() if not t.is_floating_point() else t.float()
            return t.contiguous(), int(t.dtype == torch.int64)

        # (float64 outputs -- the reference's dtype, train_model.py:106 -- are the kernels' float32 values cast up: the cast back is exact)
        key, tonic = key_out.detach().float().contiguous(), tonic_out.detach().float().contiguous()
        genre = genre_out.detach().float().contiguous() if genre_out is not None else None
        kl = key_labels.to(device=dev, dtype=torch.float32).contiguous()
        tl, tl64 = onehot(tonic_labels)
        sl, sl64 = onehot(key_signature_id)
        gl, gl64 = onehot(genre_labels) if genre is not None else (None, 0)
        need_grad = any(ctx.needs_input_grad[:3])
        scal = torch.empty(10, dtype=torch.float32, device=dev)
        grads = torch.empty((B, 35), dtype=torch.float32, device=dev) if need_grad else None
        ptr = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().ake_general_step_f32(
                key.data_ptr(), tonic.data_ptr(), ptr(genre), kl.data_ptr(), tl.data_ptr(), tl64, ptr(gl), gl64, sl.data_ptr(), sl64, B,
                weights[0], weights[1], weights[2], int(bool(use_cos)), scal.data_ptr(),
                grads.data_ptr() if need_grad else None, grads.data_ptr() + 4 * B * 12 if need_grad else None,
                grads.data_ptr() + 4 * B * 24 if need_grad and genre is not None else None, torch.cuda.current_stream().cuda_stream),
                "ake_general_step_f32")
        ctx.grads, ctx.B, ctx.has_genre, ctx.out_dtype = grads, B, genre is not None, key_out.dtype
        return scal

    @staticmethod
    def backward(ctx, g):
        B = ctx.B
        g0 = g[0]
        scaled = (ctx.grads * g0).to(ctx.out_dtype).view(-1)             # one multiply (+ one cast) for the three gradients
        d_key = scaled[:B * 12].view(B, 12)
        d_tonic = scaled[B * 12:B * 24].view(B, 12)
        d_genre = scaled[B * 24:B * 35].view(B, 11) if ctx.has_genre else None
        return d_key, d_tonic, d_genre, None, None, None, None, None, None


def _opt_get(opt, name, default):
    return getattr(opt, name, default) if opt is not None else default


class PitchClassNet(LightningModule):
    fused_loss = True     # general_step as one launch on the device (csrc/loss.hip); False: the same arithmetic as torch ops (A/B, tests)

    def __init__(self, pitches, pitch_classes, num_layers, kernel_size, opt=None, window_size=23, batch_size=4,
                 train_set=None, val_set=None):
        super().__init__()
        self.pitches, self.pitch_classes = pitches, pitch_classes
        self.num_layers, self.kernel_size = num_layers, kernel_size
        self.batch_size, self.window_size, self.opt = batch_size, window_size, opt
        self.conv_layers = _opt_get(opt, "conv_layers", 3)
        self.n_filters = _opt_get(opt, "n_filters", 4)
        self.best_mirex_score = 0
        self.data = {"train": train_set, "val": val_set}
        for flag in _VARIANT_FLAGS:
            if _opt_get(opt, flag, False):
                raise NotImplementedError(f"--{flag} selects a non-default architecture variant that the HIP path does not build "
                                          "(SURVEY.md section 2.1); only the default PitchClassNet family is available")
        nf, k = self.n_filters, kernel_size
        # --local (sliding-window key tracking, models.py:720-722): no time pooling in the layers, the key / tonic heads end in
        # MaxPool2d((1, W), stride 1) -- parameter-free, so the state_dict is the default net's (inference and training).
        self.local = bool(_opt_get(opt, "local", False))
        self.local_window = 0
        if self.local:
            self.local_window = int(_opt_get(opt, "frames", 5) * _opt_get(opt, "loc_window_size", 10)
                                    - _opt_get(opt, "head_layers", 2) * (kernel_size - 1))
            if self.local_window < 1:
                raise ValueError("--local: frames * loc_window_size must exceed head_layers * (kernel_size - 1)")
        # --resblock (models.py:181-187, 218-224, 402-454): the stacks are one conv + conv_layers residual blocks (inference and training).
        self.resblock = bool(_opt_get(opt, "resblock", False))
        self.pc2p_mem = bool(_opt_get(opt, "pc2p_mem", False))
        self.p2pc_conv = bool(_opt_get(opt, "p2pc_conv", False))
        self.stay_sixth = bool(_opt_get(opt, "stay_sixth", False))
        # --denseblock (models.py:188-189, 225-226, 456-648): DenseNet-style stacks (inference and training).
        self.denseblock = bool(_opt_get(opt, "denseblock", False))
        if self.denseblock and (self.resblock or self.pc2p_mem or self.p2pc_conv or self.stay_sixth or self.local):
            raise NotImplementedError("--denseblock together with --resblock / --pc2p_mem / --p2pc_conv / --stay_sixth / --local is not built")
        if self.stay_sixth and self.pc2p_mem:
            raise NotImplementedError("--stay_sixth together with --pc2p_mem is not built")
        self.model = nn.Sequential(*[PitchClassNetLayer(i, nf, k, self.conv_layers, self.resblock, self.pc2p_mem, self.p2pc_conv, pitches,
                                                        self.stay_sixth, self.denseblock) for i in range(num_layers)])
        final = nf if num_layers == 1 else self.model[num_layers - 1].out_pc          # models.py:694-710
        if self.denseblock:
            final = self.model[num_layers - 1].out_pc                                 # models.py:678-689
        self.head_layers = _opt_get(opt, "head_layers", 2)
        self.genre = bool(_opt_get(opt, "genre", False))
        t, kk, g = [], [], []
        ch = final
        for i in range(self.head_layers):                                              # models.py:716-737
            if i == self.head_layers - 1:
                t.append(EquivariantPitchClassConvolutionSimple(12, ch, 1, k))
                kk.append(EquivariantPitchClassConvolutionSimple(12, ch, 1, k))
                if self.genre:
                    g.append(nn.Conv2d(ch, 1, kernel_size=(2, k)))
            else:
                co = 2 * ch if i == 0 else ch
                t += [EquivariantPitchClassConvolutionSimple(12, ch, co, k), nn.BatchNorm2d(co), nn.LeakyReLU()]
                kk += [EquivariantPitchClassConvolutionSimple(12, ch, co, k), nn.BatchNorm2d(co), nn.LeakyReLU()]
                if self.genre:
                    g += [nn.Conv2d(ch, co, kernel_size=(1, k)), nn.BatchNorm2d(co), nn.LeakyReLU()]
                ch = co
        self.tonic_classifier = nn.Sequential(*t)
        self.key_classifier = nn.Sequential(*kk)
        if self.genre:
            self.genre_classifier = nn.Sequential(*g)
        self.sig = nn.Sigmoid()
        # device-side state (not part of the state_dict)
        self._h = None
        self._h_device = None
        self._h_stamp = None
        self._ws = None              # workspace of the forwards that no backward depends on (inference, train mode without autograd)
        self._ws_last = None         # workspace of the most recent forward (what `tap` reads)
        self._train_ws_free = []     # training workspaces no autograd node owns at the moment
        # opt.precision (not a reference flag): arithmetic of the INFERENCE convolutions, ake_pcnet_config::precision -- "mixed" (default: f16
        # single-product pitch / semitone / layer-0 convolutions, split-bf16 x 3 elsewhere) or "f32x3" (no operand rounded below 2^-17)
        prec = _opt_get(opt, "precision", "mixed")
        if prec not in PRECISIONS:
            raise ValueError(f"opt.precision must be one of {sorted(k for k in PRECISIONS if isinstance(k, str))}, got {prec!r}")
        self.precision = PRECISIONS[prec]
        self._flat = None            # one float32 device buffer holding every float state_dict entry (layout: ake_pcnet_grad_offset)
        self._flat_grad = None       # same layout; every p.grad is a view of it
        self._attached = False       # parameters/buffers are views of _flat
        self._dirty = 0              # bumped by the raw-pointer writers of the weights (fused Adam)
        self._stats_dirty = 0        # ... of the running statistics: only the eval-mode packs fold them, so a training forward need not repack
        self._h_stats_stamp = 0
        self._h_eval_stale = False    # the handle's inference fragments are behind its weights (last load: ake_pcnet_load_for_training_f32)

    # ------------------------------------------------------------------ device handle
    def _float_state(self):
        return [(k, v) for k, v in self.state_dict(keep_vars=True).items() if v.is_floating_point()]

    def _config(self):
        c = _lib.PcnetConfig()
        c.pitches, c.pitch_classes, c.num_layers, c.kernel_size = self.pitches, self.pitch_classes, self.num_layers, self.kernel_size
        c.conv_layers, c.n_filters, c.head_layers = self.conv_layers, self.n_filters, self.head_layers
        c.time_pool_size = _opt_get(self.opt, "time_pool_size", 2)
        c.genre = 1 if self.genre else 0
        c.max_pool = 1 if _opt_get(self.opt, "max_pool", False) else 0
        c.local = self.local_window
        c.resblock = 1 if self.resblock else 0
        c.pc2p_mem = 1 if self.pc2p_mem else 0
        c.p2pc_conv = 1 if self.p2pc_conv else 0
        c.stay_sixth = 1 if self.stay_sixth else 0
        c.denseblock = 1 if self.denseblock else 0
        c.precision = self.precision
        return c

    def _layout(self):
        """[(state_dict key, float offset, count)] of the flat parameter / gradient buffers, from the C ABI."""
        if getattr(self, "_layout_cache", None) is None or self._layout_cache[0] is not self._h:
            L = _lib.lib()
            rows = []
            for i in range(L.ake_pcnet_num_tensors(self._h)):
                name, shape, ndim = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
                _lib.check(L.ake_pcnet_tensor_info(self._h, i, C.byref(name), shape, C.byref(ndim)), "ake_pcnet_tensor_info")
                cnt = 1
                for d in range(ndim.value):
                    cnt *= shape[d]
                rows.append((name.value.decode(), int(L.ake_pcnet_grad_offset(self._h, name.value)), int(cnt)))
            self.__dict__["_layout_cache"] = (self._h, rows)
        return self._layout_cache[1]

    def _is_attached(self, state):
        if self._flat is None or not self._attached:
            return False
        base = self._flat.data_ptr()
        return all(v.dtype == torch.float32 and v.data_ptr() == base + 4 * off for (_, off, _), v in zip(self._layout(), state))

    def _sync_weights(self, device, for_eval=None):
        """Make the device handle reflect the current parameters (after an optimizer step, load_state_dict, .to()).

        ``for_eval``: the caller is about to run EVAL-mode kernels (default: ``not self.training``).  Only the eval-mode packs fold the
        running statistics, so a train-mode forward skips the repack after a statistics update -- but ``KeyEstimator`` and ``prepare()``
        serve through the eval kernels whatever ``self.training`` says, and pass True (stale BatchNorm statistics otherwise).

        float32 parameters on the device are MOVED into one flat buffer (the tensors become views of it), so that the
        packed kernel weights are rebuilt on the device (ake_pcnet_load_from_device_f32) and the fused optimizer / the
        gradient all-reduce work on one buffer.  Other dtypes (the reference trains in float64, train_model.py:106) keep
        their own storage and are cast into the flat buffer whenever they change."""
        L = _lib.lib()
        with torch.cuda.device(device):
            if self._h is None or self._h_device != device:
                self._release()
                cfg = self._config()
                h = C.c_void_p()
                _lib.check(L.ake_pcnet_create(C.byref(cfg), C.byref(h)), "ake_pcnet_create")
                self._h, self._h_device = h, device
                self._flat = self._flat_grad = None
                self._attached = False
            layout = self._layout()
            sd = dict(self._float_state())
            expected, have = {k for k, _, _ in layout}, set(sd)
            if expected != have:
                raise _lib.AkeError(f"state_dict keys differ from the device layout: missing {sorted(expected - have)[:4]}, "
                                    f"unexpected {sorted(have - expected)[:4]}")
            state = [sd[k] for k, _, _ in layout]
            total = int(L.ake_pcnet_grad_floats(self._h))
            if not self._is_attached(state):
                flat = torch.empty(total, dtype=torch.float32, device=device)
                can_attach = all(v.dtype == torch.float32 and v.device == device for v in state)
                with torch.no_grad():
                    for (_, off, cnt), v in zip(layout, state):
                        flat[off:off + cnt].copy_(v.detach().reshape(-1))
                        if can_attach:
                            v.data = flat[off:off + cnt].view(v.shape)
                self._flat, self._attached = flat, can_attach
                self._flat_grad = None
                self._h_stamp = None
            elif not self._attached:
                pass
            stamp = (self._dirty,) + tuple((v._version, v.data_ptr()) for v in state)
            # (the reference trains with 8 clips per step: repacking after every step's running-statistics update cost 0.35 ms of 3.7)
            if for_eval is None:
                for_eval = not self.training
            if stamp == self._h_stamp and (not for_eval or (self._stats_dirty == self._h_stats_stamp and not self._h_eval_stale)):
                return
            if not self._attached and self._h_stamp is not None:      # staged copy of foreign-dtype parameters
                with torch.no_grad():
                    for (_, off, cnt), v in zip(layout, state):
                        self._flat[off:off + cnt].copy_(v.detach().reshape(-1))
            # a training forward needs only the training fragments (half of the repack launches); the next eval-mode use does the full load
            load = L.ake_pcnet_load_from_device_f32 if for_eval else L.ake_pcnet_load_for_training_f32
            _lib.check(load(self._h, self._flat.data_ptr(), torch.cuda.current_stream().cuda_stream), "ake_pcnet_load_from_device_f32")
            self._h_stamp = stamp
            self._h_eval_stale = not for_eval
            if for_eval:
                self._h_stats_stamp = self._stats_dirty

    def _grads_in_place(self):
        """True when every p.grad is (or can be made) a view of the flat gradient buffer; prepares it for accumulation."""
        if not self._attached:
            return False
        params = dict(self.named_parameters())
        offs = self._grad_offsets()
        if self._flat_grad is None:
            self._flat_grad = torch.zeros_like(self._flat)
        base = self._flat_grad.data_ptr()
        none = [n for n, p in params.items() if p.grad is None]
        ours = all(p.grad is None or p.grad.data_ptr() == base + 4 * offs[n] for n, p in params.items())
        if not ours or any(not p.requires_grad for p in params.values()):
            return False
        if none:
            if len(none) == len(params):
                self._flat_grad.zero_()
            for n in none:
                p = params[n]
                view = self._flat_grad[offs[n]:offs[n] + p.numel()].view(p.shape)
                if len(none) != len(params):
                    view.zero_()
                p.grad = view
        return True

    def flat_parameters(self):
        """(flat float32 parameter buffer, flat gradient buffer or None) -- what the fused optimizer and the gradient
        all-reduce operate on.  Valid once the module is on the device (calls prepare())."""
        self.prepare()
        if self._attached and self._flat_grad is None:
            self._grads_in_place()
        return self._flat, self._flat_grad

    def mark_parameters_changed(self):
        """Call after writing the flat buffer through a raw pointer (fused optimizer): the next forward repacks the weights."""
        self._dirty += 1

    def _release(self):
        d = self.__dict__                    # plain attributes; avoids nn.Module.__setattr__ at interpreter shutdown
        h = d.get("_h")
        if h is not None:
            try:
                _lib.lib().ake_pcnet_destroy(h)
            except Exception:      # noqa: BLE001
                pass
        d["_h"] = None
        d["_h_stamp"] = None

    def __del__(self):
        try:
            self._release()
        except Exception:          # noqa: BLE001
            pass

    def _workspace(self, nbytes, device):
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            self._ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        self._ws_last = self._ws
        return self._ws

    def _train_ws_acquire(self, B, Tn, device):
        """A training workspace that no pending backward reads: the smallest free one that is large enough, else a new one."""
        nbytes = int(_lib.lib().ake_pcnet_train_workspace_bytes(self._h, B, Tn))
        fit = [w for w in self._train_ws_free if w.numel() >= nbytes and w.device == device]
        if fit:
            ws = min(fit, key=lambda w: w.numel())
            self._train_ws_free = [w for w in self._train_ws_free if w is not ws]
            return ws
        self._train_ws_free = [w for w in self._train_ws_free if w.device == device][-1:]      # drop outgrown buffers, keep at most one
        return torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)

    def _train_ws_release(self, ws):
        if len(self._train_ws_free) < 2:
            self._train_ws_free.append(ws)

    @property
    def handle(self):
        """Opaque ``ake_pcnet*`` (valid after a forward or ``prepare()``)."""
        return self._h

    def precision_dtype(self) -> str:
        """What the inference convolutions of this module's DEVICE HANDLE compute in (read back through ake_pcnet_precision, so that a
        benchmark line cannot claim a precision the handle does not run)."""
        self._sync_weights(self._device(), for_eval=True)
        return PRECISION_DTYPES[int(_lib.lib().ake_pcnet_precision(self._h))]

    def prepare(self):
        """Upload the current weights to the device now (otherwise done lazily by ``forward``)."""
        self._sync_weights(self._device(), for_eval=True)
        return self

    def _device(self):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise _lib.AkeError("PitchClassNet.forward runs on a HIP device only: move the module with .cuda() "
                                "(the reference does so in its constructors, models.py:199,237,739). No CPU fallback.")
        return dev

    # ------------------------------------------------------------------ forward (models.py:747-817)
    def forward(self, mel, seq_length):
        device = self._device()
        self._sync_weights(device)
        assert mel.dim() == 4 and mel.shape[1] == 1 and mel.shape[2] == self.pitches, \
            f"mel must be (B,1,{self.pitches},T), got {tuple(mel.shape)}"                     # models.py:357
        out_dtype = mel.dtype if mel.is_floating_point() else torch.float32
        x = mel.to(device=device, dtype=torch.float32).contiguous()
        B, _, _, Tn = x.shape
        if self.local and not self.training:
            return self._forward_local(x, out_dtype)
        seq = None
        if seq_length is not None and not self.local:          # --local: per-frame outputs, seq_length only enters the losses
            seq = torch.as_tensor(seq_length).to(device=device, dtype=torch.int64).reshape(-1)
            if seq.numel() == 1 and B > 1:
                seq = seq.expand(B)
            seq = seq.contiguous()
            assert seq.numel() == B
        key = torch.empty((B, 12), dtype=torch.float32, device=device)
        tonic = torch.empty((B, 12), dtype=torch.float32, device=device)
        genre = torch.empty((B, 11), dtype=torch.float32, device=device) if self.genre else None
        L = _lib.lib()
        ptr = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream().cuda_stream
            if self.training and self.p2pc_conv and self.stay_sixth:
                raise NotImplementedError("training a --p2pc_conv --stay_sixth net is not built on the HIP path (inference only)")
            if self.training:
                # BatchNorm with batch statistics; with autograd enabled the call becomes one autograd node whose backward
                # runs the HIP backward kernels (gradients for every parameter), as loss.backward() does in the reference
                if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                    outs = _TrainStep.apply(self, x, seq, *self.parameters())
                    key, tonic = outs[0], outs[1]
                    genre = outs[2] if self.genre else None
                else:
                    key, tonic, genre = self._forward_train_raw(x, seq)
            else:
                ws = self._workspace(L.ake_pcnet_workspace_bytes(self._h, B, Tn), device)
                _lib.check(L.ake_pcnet_forward_f32(self._h, x.data_ptr(), B, Tn, ptr(seq), key.data_ptr(), tonic.data_ptr(),
                                                   ptr(genre), ws.data_ptr(), ws.numel(), stream), "ake_pcnet_forward_f32")
        self._last_shape = (B, Tn)
        if self.genre:
            return key.to(out_dtype), tonic.to(out_dtype), genre.to(out_dtype)               # models.py:813
        return key.to(out_dtype), tonic.to(out_dtype)                                         # models.py:815

    def _forward_local(self, x, out_dtype):
        """--local (models.py:805-810): per-frame outputs key (B, T', 12) with sigmoid, tonic (B, T', 12), genre (B, Tm, 11); the
        shapes are the reference's ``reshape`` of the (B, 1, rows, frames) maps -- a reinterpretation, not a transpose."""
        device = x.device
        B, _, _, Tn = x.shape
        L = _lib.lib()
        key, tonic, genre = self._empty_outputs(B, Tn, device)
        with torch.cuda.device(device):
            ws = self._workspace(L.ake_pcnet_workspace_bytes(self._h, B, Tn), device)
            _lib.check(L.ake_pcnet_forward_local_f32(self._h, x.data_ptr(), B, Tn, key.data_ptr(), tonic.data_ptr(),
                                                     genre.data_ptr() if genre is not None else None, ws.data_ptr(), ws.numel(),
                                                     torch.cuda.current_stream().cuda_stream), "ake_pcnet_forward_local_f32")
        self._last_shape = (B, Tn)
        if self.genre:
            return key.to(out_dtype), tonic.to(out_dtype), genre.to(out_dtype)
        return key.to(out_dtype), tonic.to(out_dtype)

    def _empty_outputs(self, B, Tn, device):
        """Output tensors of one forward: (B, 12) / (B, 11), or with --local the per-frame (B, T', 12) / (B, Tm, 11)."""
        if not self.local:
            return (torch.empty((B, 12), dtype=torch.float32, device=device), torch.empty((B, 12), dtype=torch.float32, device=device),
                    torch.empty((B, 11), dtype=torch.float32, device=device) if self.genre else None)
        tq, tm = C.c_int(), C.c_int()
        _lib.check(_lib.lib().ake_pcnet_local_frames(self._h, Tn, C.byref(tq), C.byref(tm)), "ake_pcnet_local_frames")
        if tq.value < 1:
            raise _lib.AkeError(f"--local: {Tn} frames leave {tm.value} map frames, fewer than the pooling window {self.local_window}")
        return (torch.empty((B, tq.value, 12), dtype=torch.float32, device=device),
                torch.empty((B, tq.value, 12), dtype=torch.float32, device=device),
                torch.empty((B, tm.value, 11), dtype=torch.float32, device=device) if self.genre else None)

    def _forward_train_raw(self, x, seq, ws=None):
        """x (B,1,P,T) float32 contiguous on the device -> float32 outputs; updates the BatchNorm running statistics.  ``ws``: the
        workspace the activations are left in (an autograd node's own, see _TrainStep); default: the module's shared one."""
        device = x.device
        B, _, _, Tn = x.shape
        L = _lib.lib()
        key, tonic, genre = self._empty_outputs(B, Tn, device)
        ptr = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(device):
            if ws is None:
                ws = self._workspace(L.ake_pcnet_train_workspace_bytes(self._h, B, Tn), device)
            self._ws_last = ws
            n_ch = sum(c for _, c, _ in self._bn_layers())
            stats = torch.empty((n_ch, 3), dtype=torch.float32, device=device)
            _lib.check(L.ake_pcnet_forward_train_f32(self._h, x.data_ptr(), B, Tn, ptr(seq), key.data_ptr(), tonic.data_ptr(),
                                                     ptr(genre), stats.data_ptr(), ws.data_ptr(), ws.numel(),
                                                     torch.cuda.current_stream().cuda_stream), "ake_pcnet_forward_train_f32")
            if self._attached:
                _lib.check(L.ake_pcnet_update_running_stats_f32(self._h, stats.data_ptr(), self._flat.data_ptr(), 0.1,
                                                                torch.cuda.current_stream().cuda_stream), "ake_pcnet_update_running_stats_f32")
        self.__dict__["_last_bn_stats"] = stats
        if self._attached:
            self._stats_dirty += 1                             # eval-mode packs fold the running statistics: repacked at the next eval forward
            with torch.no_grad():
                torch._foreach_add_(self._nbt_tensors(), 1)
        else:
            self._update_running_stats(stats)
        return key, tonic, genre

    def _nbt_tensors(self):
        if getattr(self, "_nbt_cache", None) is None or self._nbt_cache[0] is not self._h:
            mods = dict(self.named_modules())
            self.__dict__["_nbt_cache"] = (self._h, [mods[name] for name, _, _ in self._bn_layers()])
        return [m.num_batches_tracked for m in self._nbt_cache[1]]

    _DEVICE_STATE = ("_h", "_h_device", "_h_stamp", "_ws", "_ws_last", "_flat", "_flat_grad", "_layout_cache", "_goff_cache", "_bn_cache", "_nbt_cache", "_last_bn_stats")

    def __getstate__(self):
        """copy.deepcopy / pickle: the device handle and the flat buffers belong to this object only."""
        d = self.__dict__.copy()
        for k in self._DEVICE_STATE:
            d[k] = None
        d["_attached"] = False
        d["_train_ws_free"] = []
        return d

    def _grad_offsets(self):
        if getattr(self, "_goff_cache", None) is None or self._goff_cache[0] is not self._h:
            L = _lib.lib()
            offs = {name: int(L.ake_pcnet_grad_offset(self._h, name.encode())) for name, _ in self.named_parameters()}
            assert all(v >= 0 for v in offs.values())
            self.__dict__["_goff_cache"] = (self._h, offs)
        return self._goff_cache[1]

    def _backward_raw(self, x, seq, key, d_key, d_tonic, d_genre, into=None, ws=None):
        """Flat float32 gradient buffer (state_dict order) for the train-mode forward of (x, seq) whose activations are in ``ws``
        (default: the workspace of the most recent forward); ``into`` accumulates."""
        ws = ws if ws is not None else self._ws_last
        device = x.device
        B, _, _, Tn = x.shape
        L = _lib.lib()
        f32 = lambda t: None if t is None else t.to(device=device, dtype=torch.float32).contiguous()
        d_key, d_tonic, d_genre = f32(d_key), f32(d_tonic), f32(d_genre)
        if self.genre and d_genre is None:
            d_genre = torch.zeros_like(self._empty_outputs(B, Tn, device)[2])
        flat = into if into is not None else torch.empty(int(L.ake_pcnet_grad_floats(self._h)), dtype=torch.float32, device=device)
        ptr = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(device):
            _lib.check(L.ake_pcnet_backward_f32(self._h, x.data_ptr(), B, Tn, ptr(seq), key.data_ptr(), d_key.data_ptr(), d_tonic.data_ptr(),
                                                ptr(d_genre), flat.data_ptr(), 1 if into is not None else 0, ws.data_ptr(), ws.numel(),
                                                torch.cuda.current_stream().cuda_stream), "ake_pcnet_backward_f32")
        return flat

    def _bn_layers(self):
        """[(reference module path, channels, channel offset)] of the BatchNorm layers in the device's forward order."""
        if getattr(self, "_bn_cache", None) is None or self._bn_cache[0] is not self._h:
            L, out = _lib.lib(), []
            for i in range(L.ake_pcnet_num_bn(self._h)):
                name, ch, off = C.c_char_p(), C.c_int(), C.c_int()
                _lib.check(L.ake_pcnet_bn_info(self._h, i, C.byref(name), C.byref(ch), C.byref(off)), "ake_pcnet_bn_info")
                out.append((name.value.decode(), ch.value, off.value))
            self.__dict__["_bn_cache"] = (self._h, out)
        return self._bn_cache[1]

    @torch.no_grad()
    def _update_running_stats(self, stats):
        """nn.BatchNorm2d's train-mode side effect: running stats <- momentum blend with (mean, unbiased var) of the batch."""
        mods = dict(self.named_modules())
        for name, ch, off in self._bn_layers():
            bn = mods[name]
            mean, var, cnt = stats[off:off + ch, 0], stats[off:off + ch, 1], stats[off:off + ch, 2]
            m = bn.momentum if bn.momentum is not None else 0.1
            unbiased = var * cnt / (cnt - 1).clamp_min(1)
            bn.running_mean.mul_(1 - m).add_(mean.to(bn.running_mean.dtype) * m)
            bn.running_var.mul_(1 - m).add_(unbiased.to(bn.running_var.dtype) * m)
            bn.num_batches_tracked += 1
        # the device copy of the weights does not depend on the running statistics in train mode; eval re-syncs lazily

    @torch.no_grad()
    def _update_recomputed_running_stats(self, stats):
        """--denseblock: the reference checkpoints norm1 + conv1 of every dense layer (models.py:484-489, 553); autograd's backward runs that
        half again in train mode, so those BatchNorm layers blend the batch statistics a second time and count two batches per step."""
        if self._attached:
            with torch.cuda.device(self._h_device):
                _lib.check(_lib.lib().ake_pcnet_update_recomputed_running_stats_f32(self._h, stats.data_ptr(), self._flat.data_ptr(), 0.1, None,
                                                                                     torch.cuda.current_stream().cuda_stream),
                           "ake_pcnet_update_recomputed_running_stats_f32")
            self._stats_dirty += 1
        mods = dict(self.named_modules())
        for name, ch, off in self._bn_layers():
            if not name.endswith(".norm1"):
                continue
            bn = mods[name]
            if not self._attached:
                mean, var, cnt = stats[off:off + ch, 0], stats[off:off + ch, 1], stats[off:off + ch, 2]
                m = bn.momentum if bn.momentum is not None else 0.1
                bn.running_mean.mul_(1 - m).add_(mean.to(bn.running_mean.dtype) * m)
                bn.running_var.mul_(1 - m).add_((var * cnt / (cnt - 1).clamp_min(1)).to(bn.running_var.dtype) * m)
            bn.num_batches_tracked += 1

    @staticmethod
    def keep_taps(on=True):
        """Debug: keep every activation that `tap` can name in memory (inference otherwise fuses the semitone conv into the last
        pitch conv of a stack and runs the last pitch-class stack as one launch: `model.i.p2p.layer.8` and the last layer's
        `pc2pc.layer.*` are then never written).  Process-wide; returns the previous setting."""
        return bool(_lib.lib().ake_debug_keep_taps(1 if on else 0))

    def tap(self, name):
        """Intermediate activation of the last forward (debug / bisecting): reference module path -> tensor."""
        B, Tn = self._last_shape
        L = _lib.lib()
        shape = (C.c_int64 * 4)()
        _lib.check(L.ake_pcnet_tap_info(self._h, name.encode(), B, Tn, shape), "ake_pcnet_tap_info")
        out = torch.empty(tuple(shape), dtype=torch.float32, device=self._h_device)
        with torch.cuda.device(self._h_device):
            _lib.check(L.ake_pcnet_tap_copy(self._h, name.encode(), B, Tn, self._ws_last.data_ptr(), out.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream), "ake_pcnet_tap_copy")
        return out

    # ------------------------------------------------------------------ steps (models.py:819-1027)
    def _general_step_local(self, batch):
        """--local branch of general_step (models.py:861-876, 898-909): per-frame labels (B, T', 12); for every clip the losses and
        scores cover its first ``seq_length - loc_window_size * frames + 1`` frames and are averaged over the batch.  The tonic accuracy
        covers two frames fewer (the reference's ``seq_length - (loc_window_size * frames + 1)``, :906 -- kept).  The reference's
        --local --genre lines (:866-873, :907-911) index a (B, 11) "mask" into per-frame tensors and re-mask inside the clip loop;
        they cannot run for any batch, so that combination is refused here rather than invented."""
        opt = self.opt
        if self.genre:
            raise NotImplementedError("general_step with --local and --genre: the reference's lines (models.py:866-873) do not run; "
                                      "forward / backward of such a net are available, the loss is the caller's")
        mel = batch["mel"]
        out = self.forward(mel, None)
        key_out, tonic_out = out[0], out[1]
        dev = key_out.device
        key_labels = batch["key_labels"].to(device=dev, dtype=key_out.dtype)
        tonic_labels = batch["tonic_labels"].long().to(dev)
        tonic_idx = torch.argmax(tonic_labels, dim=2)
        key_signature_id = batch["key_signature_id"].to(dev)
        span = _opt_get(opt, "loc_window_size", 10) * _opt_get(opt, "frames", 5)
        seq = [int(v) for v in torch.as_tensor(batch["seq_length"]).reshape(-1).tolist()]
        B = mel.shape[0]
        bce = tonic_loss = 0
        sums = [0.0] * 7
        acc_tonic = 0.0
        for i in range(B):
            n = seq[i] - span + 1
            bce = bce + F.binary_cross_entropy(key_out[i, :n], key_labels[i, :n])
            tonic_loss = tonic_loss + F.cross_entropy(tonic_out[i, :n], tonic_idx[i, :n])
            with torch.no_grad():
                sub = self.mirex_score(key_labels[i, :n], key_out[i, :n], tonic_labels[i, :n], tonic_out[i, :n], key_signature_id[i, :n])
                sums = [a + b for a, b in zip(sums, sub)]
                m = seq[i] - (span + 1)
                acc_tonic = acc_tonic + (torch.argmax(tonic_out[i, :m], dim=1) == tonic_idx[i, :m]).float().mean()
        loss = _opt_get(opt, "key_weight", 1.0) * bce / B + _opt_get(opt, "tonic_weight", 1.0) * tonic_loss / B
        if _opt_get(opt, "use_cos", False):
            loss = loss + (1 - F.cosine_similarity(key_out, key_labels, dim=1).sum() / key_out.shape[0])
        mirex, correct, fifths, relative, parallel, other, accuracy = [torch.as_tensor(v / B).clone().float() for v in sums]
        return (loss, accuracy, mirex, correct, fifths, relative, parallel, other, torch.as_tensor(acc_tonic / B).float(),
                torch.tensor(0.0))

    def general_step(self, batch, batch_idx, mode):
        opt = self.opt
        if self.local:
            return self._general_step_local(batch)
        mel = batch["mel"]
        key_signature_id = batch["key_signature_id"]
        out = self.forward(mel, batch["seq_length"] if _opt_get(opt, "frames", 5) > 0 else None)
        key_out, tonic_out = out[0], out[1]
        dev = key_out.device
        if (key_out.is_cuda and key_out.dtype in (torch.float32, torch.float64) and type(self).mirex_score is PitchClassNet.mirex_score
                and self.fused_loss):
            # the device path: one launch for the loss, its gradient and the metrics (a subclass that overrides mirex_score keeps the torch ops)
            weights = (float(_opt_get(opt, "key_weight", 1.0)), float(_opt_get(opt, "tonic_weight", 1.0)), float(_opt_get(opt, "genre_weight", 0.1)))
            vals = _FusedGeneralStep.apply(key_out, tonic_out, out[2] if self.genre else None, batch["key_labels"], batch["tonic_labels"],
                                           batch["genre"] if self.genre else None, key_signature_id, weights, _opt_get(opt, "use_cos", False))
            return tuple(vals.unbind(0))
        key_labels = batch["key_labels"].to(mel.dtype if mel.is_floating_point() else torch.float32)
        tonic_labels = batch["tonic_labels"].long()
        tonic_idx = torch.argmax(tonic_labels, dim=1)
        if self.genre:
            genre_labels = batch["genre"].long()
            genre_idx = torch.argmax(genre_labels, dim=1)
            genre_mask = genre_labels.sum(dim=1) == 1                                        # models.py:839
        key_labels, tonic_idx = key_labels.to(dev), tonic_idx.to(dev)
        loss = _opt_get(opt, "key_weight", 1.0) * F.binary_cross_entropy(key_out, key_labels.to(key_out.dtype)) \
            + _opt_get(opt, "tonic_weight", 1.0) * F.cross_entropy(tonic_out, tonic_idx)      # models.py:878-889
        accuracy_genre = torch.tensor(0.0)
        if self.genre:
            genre_mask, genre_idx = genre_mask.to(dev), genre_idx.to(dev)
            # models.py:892-893: `if genre_mask.sum() != 0: CE(genre_out[mask], labels[mask])` -- the same value without the host
            # round trip (a device -> host sync in the middle of every step left the GPU idle for milliseconds): masked mean of the
            # per-sample losses, an exact zero when no clip of the batch carries a genre label
            m = genre_mask.to(out[2].dtype)
            cnt = m.sum()
            per = F.cross_entropy(out[2], genre_idx, reduction="none")
            loss = loss + _opt_get(opt, "genre_weight", 0.1) * ((per * m).sum() / cnt.clamp(min=1.0))
            accuracy_genre = (((torch.argmax(out[2], dim=1) == genre_idx).to(m.dtype) * m).sum() / cnt.clamp(min=1.0)).float()
        if _opt_get(opt, "use_cos", False):                                                   # models.py:885-896
            loss = loss + (1 - F.cosine_similarity(key_out, key_labels.to(key_out.dtype), dim=1).sum() / key_out.shape[0])
        mirex, correct, fifths, relative, parallel, other, accuracy = self.mirex_score(
            key_labels, key_out, tonic_labels.to(dev), tonic_out, key_signature_id.to(dev))
        accuracy_tonic = (torch.argmax(tonic_out, dim=1) == tonic_idx).float().mean()
        return loss, accuracy, mirex, correct, fifths, relative, parallel, other, accuracy_tonic, accuracy_genre

    def mirex_score(self, key_labels, key_preds, tonic_labels, tonic_preds, key_signature_id):
        return _mirex_score(key_labels, key_preds, tonic_labels, tonic_preds, key_signature_id)

    _NAMES = ("accuracy", "mirex_score", "correct", "fifths", "relative", "parallel", "other", "accuracy_tonic", "accuracy_genre")

    def _step(self, batch, batch_idx, mode):
        vals = self.general_step(batch, batch_idx, mode)
        d = {(mode + "_loss") if mode != "train" else "loss": vals[0]}
        d.update({f"{mode}_{n}": v for n, v in zip(self._NAMES, vals[1:])})
        return d

    def training_step(self, batch, batch_idx):
        d = self._step(batch, batch_idx, "train")
        d["log"] = {"loss": d["loss"]}
        acc = getattr(self.trainer, "accumulate_grad_batches", 1) if self.trainer is not None else 1
        if batch_idx % acc == 0 and (self.global_step + 1) % _opt_get(self.opt, "acc_grad", 8) == 0 and self.logger is not None:
            self.logger.experiment.add_scalar("train_loss", d["loss"], self.global_step)              # models.py:955-961
        return d

    def validation_step(self, batch, batch_idx):
        return self._step(batch, batch_idx, "val")

    def test_step(self, batch, batch_idx):
        return self._step(batch, batch_idx, "test")

    def general_end(self, outputs, mode):
        names = ("loss",) + self._NAMES
        return tuple(torch.stack([torch.as_tensor(x[f"{mode}_{n}"]).float().cpu() for x in outputs]).mean() for n in names)

    def validation_epoch_end(self, outputs):
        vals = self.general_end(outputs, "val")
        names = ("val_loss", "val_accuracy", "val_mirex_score", "val_correct", "val_fifths", "val_relative", "val_parallel",
                 "val_other", "val_accuracy_tonic", "val_accuracy_genre")
        result = dict(zip(names, vals))
        print("Val-Loss={}".format(result["val_loss"]))
        print("Val-Acc={}".format(result["val_accuracy"]))
        print("Val-Acc_Tonic={}".format(result["val_accuracy_tonic"]))
        if self.genre:
            print("Val-Acc_Genre={}".format(result["val_accuracy_genre"]))
        print("Val-Mirex_Score={}".format(result["val_mirex_score"]))
        for k, v in result.items():
            self.log(k, v)
        no_ckpt = _opt_get(self.opt, "no_ckpt", True)
        if result["val_mirex_score"] > self.best_mirex_score and not no_ckpt:                 # models.py:991-993
            self.best_mirex_score = result["val_mirex_score"]
            import os
            path = "Model_logs/lightning_logs/version_" + str(getattr(self.logger, "version", 0))
            os.makedirs(path, exist_ok=True)
            torch.save(self.state_dict(), path + "/best_model.pt")
        result["log"] = dict(result)
        return result

    def train_dataloader(self):
        return torch.utils.data.DataLoader(self.data["train"], shuffle=True, batch_size=self.batch_size)

    def val_dataloader(self):
        return torch.utils.data.DataLoader(self.data["val"], shuffle=False, batch_size=self.batch_size, drop_last=True)

    def configure_optimizers(self):
        """models.py:1017-1027.  On the device the Adam update is one fused HIP kernel over the flat parameter buffer
        (``FusedAdam`` subclasses torch.optim.Optimizer: param_groups / lr schedulers behave as with torch.optim.Adam);
        a module that is still on the CPU gets torch.optim.Adam itself."""
        lr, reg = _opt_get(self.opt, "lr", 3e-4), _opt_get(self.opt, "reg", 0)
        p0 = next(self.parameters())
        if p0.device.type == "cuda" and all(p.dtype == torch.float32 for p in self.parameters()):
            from .optim import FusedAdam
            optim = FusedAdam(self, betas=(0.9, 0.999), lr=lr, weight_decay=reg)
        else:
            optim = torch.optim.Adam(self.parameters(), betas=(0.9, 0.999), lr=lr, weight_decay=reg)
        scheduler = torch.optim.lr_scheduler.ExponentialLR(optim, gamma=_opt_get(self.opt, "gamma", 0.96))
        return [optim], [scheduler]
