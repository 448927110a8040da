"""CPU restatement of the audio-preparation stage (csrc/audio.hip).  TEST INFRASTRUCTURE ONLY.

The reference has no such stage (it takes ``waveform[0]`` at the file's own rate, KeyDataset.py:479-485); the device kernel
implements ``scipy.signal.resample_poly(x, up, down)`` with scipy's default filter.  scipy IS present in this image, so this
restatement -- the polyphase sum written out, the form the kernel evaluates -- is pinned on scipy itself
(tests/test_oracle_misc.py), and the GPU is then compared with it.
"""
from __future__ import annotations

import math

import numpy as np


def resample_filter(up: int, down: int):
    """``firwin(2 * half + 1, 1 / max(up, down), window=("kaiser", 5.0)) * up`` in float64, written out (scipy.signal.firwin:
    windowed ideal low-pass, unit gain at DC)."""
    mr = max(up, down)
    half = 10 * mr
    n = np.arange(2 * half + 1, dtype=np.float64) - half
    fc = 1.0 / mr
    h = fc * np.sinc(fc * n) * np.kaiser(2 * half + 1, 5.0)
    return h / h.sum() * up, half


def resample_poly(x, rate_in: int, rate_out: int):
    """y[k] = sum_i x[i] h[k * down - i * up + half], k < ceil(len(x) * up / down)."""
    x = np.asarray(x, np.float64)
    g = math.gcd(rate_in, rate_out)
    up, down = rate_out // g, rate_in // g
    if up == down:
        return x.copy()
    h, half = resample_filter(up, down)
    n = len(x)
    n_out = (n * up + down - 1) // down
    y = np.zeros(n_out)
    for k in range(n_out):
        t = k * down
        i_lo = 0 if t - half <= 0 else -((half - t) // up)
        i_hi = min(n - 1, (t + half) // up)
        if i_hi < i_lo:
            continue
        i = np.arange(i_lo, i_hi + 1)
        y[k] = np.dot(x[i], h[t - i * up + half])
    return y


def prepare(x, rate_in, rate_out, channel=0):
    """(C, n) -> mono (n_out,): channel selection (>= 0) or mean over the channels (-1), then resampling."""
    x = np.asarray(x, np.float64)
    mono = x[channel] if channel >= 0 else x.mean(axis=0)
    return resample_poly(mono, rate_in, rate_out)
