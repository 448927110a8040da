"""CPU restatement of the CQT front end.  TEST INFRASTRUCTURE ONLY.  **PARITY UNPINNED.**

The reference computes its input features with one third-party call,

    librosa.cqt(y, sr, hop_length=round(sr/frames), bins_per_octave=36, n_bins=36*octaves)
    -> abs -> log(1 + .)                      KeyDataset.py:485, 490-499, 505-509

librosa (pinned ``librosa-0.9.2`` at requirements.txt:250) is neither vendored
under /root/reference nor installable here, and the reference holds no golden
vector at this boundary, so this file cannot be checked against the real thing.
It restates librosa's *published definition* of the transform -- the direct-form
constant-Q transform that librosa's octave-recursive FFT implementation
approximates (up to its 1 % basis sparsification and its resampler):

    f_k   = fmin * 2^(k/bpo),  fmin = C1 = 32.70319566 Hz        (librosa default)
    N_k   = Q * sr / f_k,      Q = 1/alpha, alpha = (r^2-1)/(r^2+1), r = 2^(1/bpo)
    n     = arange(floor(-N_k/2), floor(N_k/2))                  (integer offsets)
    w     = periodic Hann of length len(n)
    C[k,t]= sqrt(N_k) * sum_n y[t*hop + n] * w[n] * exp(-2 pi i f_k n / sr) / sum(w)
    t     = 0 .. n_samples // hop     (centered frames, zero padding outside the clip)

``q_mode='librosa09'`` switches to Q = 1/(2^(1/bpo) - 1), the <=0.9.x constant.
The default follows >=0.10 because the reference's default hop (4410) is not a
multiple of 2^(octaves-1), which <=0.9.x rejects.

This is the build's own specification of the stage: the HIP multirate kernel is
tested against it, and the parity claim for the reference starts at the log-CQT
tensor (``pcnet_oracle``), not at the waveform.
"""
from __future__ import annotations

import math

import numpy as np

C1_HZ = 32.70319566257483


def cqt_frequencies(n_bins=288, bins_per_octave=36, fmin=C1_HZ):
    return fmin * 2.0 ** (np.arange(n_bins, dtype=np.float64) / bins_per_octave)


def cqt_q(bins_per_octave=36, q_mode="librosa010"):
    r = 2.0 ** (1.0 / bins_per_octave)
    if q_mode == "librosa010":
        return (r * r + 1.0) / (r * r - 1.0)
    if q_mode == "librosa09":
        return 1.0 / (r - 1.0)
    raise ValueError(q_mode)


def cqt_lengths(sr, n_bins=288, bins_per_octave=36, fmin=C1_HZ, q_mode="librosa010"):
    return cqt_q(bins_per_octave, q_mode) * sr / cqt_frequencies(n_bins, bins_per_octave, fmin)


def hop_for(sr, frames=5):
    """KeyDataset.py:485 -- ``round(rate / frames)``."""
    return int(round(sr / (frames if frames > 0 else 1)))


def n_frames(n_samples, hop):
    return 1 + n_samples // hop


def filter_taps(k_len: float):
    """Integer tap offsets and periodic-Hann window of one bin (see module docstring)."""
    lo = math.floor(-k_len / 2.0)
    hi = math.floor(k_len / 2.0)
    n = np.arange(lo, hi, dtype=np.float64)
    w = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(len(n), dtype=np.float64) / len(n))
    return n, w


def cqt_complex(y, sr, hop, n_bins=288, bins_per_octave=36, fmin=C1_HZ, q_mode="librosa010"):
    """Direct-form CQT of one clip in float64 -> complex128 (n_bins, T)."""
    y = np.asarray(y, dtype=np.float64)
    freqs = cqt_frequencies(n_bins, bins_per_octave, fmin)
    lengths = cqt_lengths(sr, n_bins, bins_per_octave, fmin, q_mode)
    T = n_frames(len(y), hop)
    pad = int(math.ceil(lengths.max() / 2.0)) + 2
    yp = np.concatenate([np.zeros(pad), y, np.zeros(pad + hop)])
    centers = pad + hop * np.arange(T)
    out = np.zeros((n_bins, T), dtype=np.complex128)
    for k in range(n_bins):
        n, w = filter_taps(lengths[k])
        kern = w * np.exp(-2j * np.pi * freqs[k] * n / sr) / w.sum()
        idx = centers[:, None] + n.astype(np.int64)[None, :]
        out[k] = math.sqrt(lengths[k]) * (yp[idx] @ kern)
    return out


def cqt_logmag(y, sr, hop, n_bins=288, bins_per_octave=36, fmin=C1_HZ, q_mode="librosa010"):
    """``log(1 + |CQT|)`` -- KeyDataset.py:497-499 -- float64 (n_bins, T)."""
    return np.log1p(np.abs(cqt_complex(y, sr, hop, n_bins, bins_per_octave, fmin, q_mode)))


class FastDirectCQT:
    """The same direct form evaluated as one dense matmul per octave (torch CPU).

    Used as the timed CPU baseline (``bench.py`` cpu_baseline leg) and checked
    against :func:`cqt_logmag` in ``tests/test_oracle_cqt.py``.  Zero-padded filters
    of one octave share a frame matrix, so the work is BLAS-shaped and uses all
    host threads.
    """

    def __init__(self, sr, hop, n_bins=288, bins_per_octave=36, fmin=C1_HZ, q_mode="librosa010", dtype=None):
        import torch
        self.torch = torch
        self.dtype = dtype or torch.float32
        self.sr, self.hop, self.n_bins, self.bpo = sr, hop, n_bins, bins_per_octave
        freqs = cqt_frequencies(n_bins, bins_per_octave, fmin)
        lengths = cqt_lengths(sr, n_bins, bins_per_octave, fmin, q_mode)
        self.banks = []
        for o0 in range(0, n_bins, bins_per_octave):
            ks = range(o0, min(o0 + bins_per_octave, n_bins))
            lo = min(math.floor(-lengths[k] / 2.0) for k in ks)
            hi = max(math.floor(lengths[k] / 2.0) for k in ks)
            W = np.zeros((hi - lo, 2 * len(ks)), dtype=np.float64)
            for j, k in enumerate(ks):
                n, w = filter_taps(lengths[k])
                kern = math.sqrt(lengths[k]) * w * np.exp(-2j * np.pi * freqs[k] * n / sr) / w.sum()
                a = int(n[0]) - lo
                W[a:a + len(n), 2 * j] = kern.real
                W[a:a + len(n), 2 * j + 1] = kern.imag
            self.banks.append((lo, hi, torch.from_numpy(W).to(self.dtype)))
        self.pad = max(-b[0] for b in self.banks) + 2

    def __call__(self, y):
        """y: (B, n) tensor/array -> (B, n_bins, T) log-magnitude."""
        torch = self.torch
        y = torch.as_tensor(y).to(self.dtype)
        B, n = y.shape
        T = n_frames(n, self.hop)
        yp = torch.nn.functional.pad(y, (self.pad, self.pad + self.hop))
        out = torch.empty((B, self.n_bins, T), dtype=self.dtype)
        for o, (lo, hi, W) in enumerate(self.banks):
            L = hi - lo
            start = self.pad + lo
            # frames[b, t, :] = yp[b, start + t*hop : start + t*hop + L]
            frames = yp.as_strided((B, T, L), (yp.stride(0), self.hop, 1), start)
            r = frames.reshape(B * T, L) @ W                       # (B*T, 2*nb)
            r = r.reshape(B, T, -1, 2)
            mag = torch.sqrt(r[..., 0] ** 2 + r[..., 1] ** 2)      # (B, T, nb)
            k0 = o * self.bpo
            out[:, k0:k0 + mag.shape[2], :] = torch.log1p(mag).transpose(1, 2)
        return out
