"""Host side of the device CQT front end.

Mirrors the one call the reference makes (KeyDataset.py:490-499):

    spec = librosa.cqt(y=waveform.numpy(), sr=rate, hop_length=hop, bins_per_octave=36, n_bins=36*octaves)
    mel  = torch.log(1 + torch.abs(torch.tensor(spec)))

as ``cqt_logmag(waveform, sr=rate, hop_length=hop, bins_per_octave=36, n_bins=36*octaves)`` on the
GPU.  torch is used for device memory and streams only; the transform itself is
``ake_cqt_logmag_f32`` (csrc/cqt.hip).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def hop_for(sample_rate: int, frames: int) -> int:
    """KeyDataset.py:485 -- ``round(rate / (opt.frames if opt.frames > 0 else 1))``."""
    return int(round(sample_rate / (frames if frames > 0 else 1)))


class CQTPlan:
    """Filter tables for one (sample rate, hop, bins) on one device; reusable across calls."""

    def __init__(self, sr: int, hop_length: int, n_bins: int = 288, bins_per_octave: int = 36, fmin: float = 0.0,
                 q_mode: int = 0, device=None, engine: int = 0):
        if not torch.cuda.is_available():
            raise _lib.AkeError("the CQT front end needs a HIP device; there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.sr, self.hop_length, self.n_bins, self.bins_per_octave = int(sr), int(hop_length), int(n_bins), int(bins_per_octave)
        cfg = _lib.CqtConfig(int(sr), int(hop_length), int(n_bins), int(bins_per_octave), float(fmin or 0.0), int(q_mode), 0, 0.0, int(engine))
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().ake_cqt_plan_create(C.byref(cfg), C.byref(self._h)), "ake_cqt_plan_create")
        self._ws = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().ake_cqt_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def num_frames(self, n_samples: int) -> int:
        return int(_lib.lib().ake_cqt_num_frames(self._h, int(n_samples)))

    def _workspace(self, nbytes: int):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
        return self._ws

    def logmag(self, audio: torch.Tensor, out_frames: int | None = None, out: torch.Tensor | None = None,
               lengths: torch.Tensor | None = None) -> torch.Tensor:
        """audio (B, n) or (n,) -> log(1+|CQT|) float32 (B, n_bins, out_frames); frames past the clip are zero.

        ``lengths`` (B,) int64: ragged batch -- row i holds ``lengths[i] <= n`` samples (the rest of the row is ignored); clip i
        gets ``1 + lengths[i] // hop`` frames and zeros after them, as ``KeyDataset.__getitem__`` pads (KeyDataset.py:245)."""
        squeeze = audio.dim() == 1
        if squeeze:
            audio = audio[None]
        audio = audio.to(device=self.device, dtype=torch.float32)
        if audio.stride(-1) != 1:
            audio = audio.contiguous()
        B, n = audio.shape
        T = self.num_frames(n)
        out_frames = T if out_frames is None else int(out_frames)
        if out is None:
            out = torch.empty((B, self.n_bins, out_frames), dtype=torch.float32, device=self.device)
        assert out.is_contiguous() and out.shape == (B, self.n_bins, out_frames) and out.dtype == torch.float32
        L = _lib.lib()
        nbytes = L.ake_cqt_workspace_bytes(self._h, B, n)
        ws = self._workspace(nbytes)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            if lengths is None:
                _lib.check(L.ake_cqt_logmag_f32(self._h, audio.data_ptr(), B, n, audio.stride(0), out.data_ptr(), out_frames,
                                                ws.data_ptr(), ws.numel(), stream), "ake_cqt_logmag_f32")
            else:
                lengths = torch.as_tensor(lengths).to(device=self.device, dtype=torch.int64).contiguous()
                assert lengths.shape == (B,)
                _lib.check(L.ake_cqt_logmag_ragged_f32(self._h, audio.data_ptr(), B, n, audio.stride(0), lengths.data_ptr(), out.data_ptr(),
                                                       out_frames, ws.data_ptr(), ws.numel(), stream), "ake_cqt_logmag_ragged_f32")
        return out[0] if squeeze else out


_plans = {}


def get_plan(sr, hop_length, n_bins=288, bins_per_octave=36, device=None, q_mode: int = 0) -> CQTPlan:
    """Cached plan.  ``q_mode``: 0 = the filter Q of librosa >= 0.10, (r^2 + 1) / (r^2 - 1) with r = 2^(1 / bins_per_octave) -- the
    build's default, because librosa <= 0.9 rejects the reference's default hop 4410; 1 = librosa <= 0.9's Q = 1 / (r - 1)
    (requirements.txt:250 pins 0.9.2: a checkpoint trained on features from that version saw this Q; its reflect padding of the
    clip edges is NOT built -- frames whose windows reach past the clip see zeros).  ADVICE r1: the choice is the caller's."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = (int(sr), int(hop_length), int(n_bins), int(bins_per_octave), str(dev), int(q_mode))
    if key not in _plans:
        _plans[key] = CQTPlan(sr, hop_length, n_bins, bins_per_octave, q_mode=q_mode, device=dev)
    return _plans[key]


def cqt_logmag(y, sr=22050, hop_length=512, n_bins=84, bins_per_octave=12, device=None, q_mode: int = 0) -> torch.Tensor:
    """``log(1 + |librosa.cqt(y, sr, hop_length, n_bins=..., bins_per_octave=...)|)`` on the GPU (defaults as librosa's); ``q_mode``
    as in get_plan."""
    y = torch.as_tensor(y)
    plan = get_plan(sr, hop_length, n_bins, bins_per_octave, device if device is not None else (y.device if y.is_cuda else None), q_mode)
    return plan.logmag(y)
