"""Device-side audio preparation in front of the CQT: channel selection / mono mix-down and polyphase resampling.

Not part of the reference, which takes channel 0 of ``torchaudio.load``'s output at the file's own sample rate
(KeyDataset.py:479-485): ``prepare(..., channel=0)`` on audio that already has the target rate is exactly that.  The rest is for
serving pipelines that hold decoded multi-channel audio of mixed rates on the GPU and want one CQT plan.
Host wrapper of ``ake_resample_f32`` (csrc/audio.hip) = ``scipy.signal.resample_poly`` with its default filter.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class Resampler:
    """``rate_in`` -> ``rate_out`` polyphase filter on one device; reusable across calls."""

    def __init__(self, rate_in: int, rate_out: int, device=None):
        if not torch.cuda.is_available():
            raise _lib.AkeError("the resampler needs a HIP device; there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.rate_in, self.rate_out = int(rate_in), int(rate_out)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ake_resampler_create(self.rate_in, self.rate_out, C.byref(self._h)), "ake_resampler_create")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().ake_resampler_destroy(self._h)
                self._h = None
        except Exception:      # noqa: BLE001
            pass

    def out_len(self, n_in: int) -> int:
        return int(_lib.lib().ake_resampler_out_len(self._h, int(n_in)))

    def __call__(self, audio: torch.Tensor, channel: int = 0, lengths: torch.Tensor | None = None):
        """audio (B, C, n) or (B, n) float32 -> (mono (B, n_out) float32, lengths_out (B,) int64).

        ``channel`` >= 0 selects that channel (0 = the reference's ``waveform[0]``), -1 takes the mean over the channels.
        ``lengths`` (B,) int64: ragged batch, clip i holds ``lengths[i] <= n`` samples; its output is zero behind its own end."""
        if audio.dim() == 2:
            audio = audio[:, None, :]
        audio = audio.to(device=self.device, dtype=torch.float32)
        if audio.stride(-1) != 1:
            audio = audio.contiguous()
        B, Cn, n = audio.shape
        n_out = self.out_len(n)
        out = torch.empty((B, n_out), dtype=torch.float32, device=self.device)
        len_out = torch.empty((B,), dtype=torch.int64, device=self.device)
        if lengths is not None:
            lengths = torch.as_tensor(lengths).to(device=self.device, dtype=torch.int64).contiguous()
            assert lengths.shape == (B,)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ake_resample_f32(self._h, audio.data_ptr(), B, Cn, n, audio.stride(0), audio.stride(1), int(channel),
                                                   lengths.data_ptr() if lengths is not None else None, out.data_ptr(), out.stride(0),
                                                   len_out.data_ptr(), torch.cuda.current_stream().cuda_stream), "ake_resample_f32")
        return out, len_out


_resamplers = {}


def get_resampler(rate_in, rate_out, device=None) -> Resampler:
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = (int(rate_in), int(rate_out), str(dev))
    if key not in _resamplers:
        _resamplers[key] = Resampler(rate_in, rate_out, dev)
    return _resamplers[key]


def prepare(audio: torch.Tensor, rate_in: int, rate_out: int, channel: int = 0, lengths=None, device=None):
    """One call: (B, C, n) at ``rate_in`` -> mono (B, n_out) at ``rate_out`` + per-clip lengths, on the GPU."""
    dev = device if device is not None else (audio.device if audio.is_cuda else None)
    return get_resampler(rate_in, rate_out, dev)(audio, channel=channel, lengths=lengths)
