"""Seeded synthetic "sine-mix" clips with key labels.

Stands in for the audio files + annotation files the reference's dataset loaders
read (KeyDataset.py:514-1233, out of scope): there is no dataset on the GPU box.
Labels follow the reference's encoding (KeyDataset.py:443-454): ``key_labels`` is
the 12-vector pitch-class set of the key signature, ``tonic_labels`` one-hot 12,
``key_signature_id`` one-hot 24 with 0..11 = C..B minor and 12..23 = C..B major
(``signature`` ordering, KeyDataset.py:524-527), ``genre`` one-hot 11.
"""
from __future__ import annotations

import numpy as np

SR = 22050
CLIP_SECONDS = 15
N_SAMPLES = SR * CLIP_SECONDS          # 330 750
_MAJOR = (0, 2, 4, 5, 7, 9, 11)
N_GENRES = 11
N_PARTIALS = 12


def key_pitch_classes(sig_id: int) -> np.ndarray:
    """Pitch-class set (12,) of key ``sig_id`` (natural minor shares its relative major's set)."""
    tonic = sig_id % 12
    major_tonic = tonic if sig_id >= 12 else (tonic + 3) % 12
    v = np.zeros(12, dtype=np.float32)
    for s in _MAJOR:
        v[(major_tonic + s) % 12] = 1.0
    return v


def clip_recipe(i: int):
    """Partials (freq, amp, phase) and labels of clip ``i``; seed = 1234 + i."""
    rng = np.random.default_rng(1234 + i)
    sig = i % 24
    tonic = sig % 12
    scale = np.flatnonzero(key_pitch_classes(sig))
    weights = np.where(scale == tonic, 3.0, 1.0)
    weights /= weights.sum()
    pcs = rng.choice(scale, size=N_PARTIALS, p=weights)
    octaves = rng.integers(2, 7, size=N_PARTIALS)
    midi = 12 * (octaves + 1) + pcs
    freq = 440.0 * 2.0 ** ((midi - 69) / 12.0)
    amp = rng.uniform(0.05, 0.25, size=N_PARTIALS)
    phase = rng.uniform(0.0, 2 * np.pi, size=N_PARTIALS)
    noise_seed = int(rng.integers(0, 2 ** 31 - 1))
    labels = {
        "key_labels": key_pitch_classes(sig),
        "tonic_labels": np.eye(12, dtype=np.float32)[tonic],
        "key_signature_id": np.eye(24, dtype=np.float32)[sig],
        "genre": np.eye(N_GENRES, dtype=np.float32)[i % N_GENRES],
    }
    return freq, amp, phase, noise_seed, labels


def make_clip(i: int, n_samples: int = N_SAMPLES, sr: int = SR):
    """float32 waveform (n_samples,) in [-0.9, 0.9] and the label dict of clip ``i``."""
    freq, amp, phase, noise_seed, labels = clip_recipe(i)
    n = np.arange(n_samples, dtype=np.float64)
    y = np.zeros(n_samples, dtype=np.float64)
    for f, a, p in zip(freq, amp, phase):
        y += a * np.sin(2 * np.pi * f * n / sr + p)
    y += np.random.default_rng(noise_seed).normal(0.0, 0.003, size=n_samples)
    y *= 0.9 / np.max(np.abs(y))
    return y.astype(np.float32), labels


def make_batch(indices, n_samples: int = N_SAMPLES, sr: int = SR):
    """Stack clips: waveforms (B, n) float32 + dict of stacked label arrays."""
    ys, labs = zip(*(make_clip(int(i), n_samples, sr) for i in indices))
    labels = {k: np.stack([l[k] for l in labs]) for k in labs[0]}
    return np.stack(ys), labels


def make_batch_device(indices, device, n_samples: int = N_SAMPLES, sr: int = SR):
    """Same recipe synthesised with torch on ``device`` (fast path for large benches).

    Partials are identical to :func:`make_clip`; the noise stream is torch's, so the
    waveforms agree with the numpy version only up to the noise floor (sigma 0.003).
    """
    import torch
    recipes = [clip_recipe(int(i)) for i in indices]
    f = torch.tensor(np.stack([r[0] for r in recipes]), device=device, dtype=torch.float64)
    a = torch.tensor(np.stack([r[1] for r in recipes]), device=device, dtype=torch.float32)
    p = torch.tensor(np.stack([r[2] for r in recipes]), device=device, dtype=torch.float64)
    out = torch.empty((len(recipes), n_samples), device=device, dtype=torch.float32)
    n = torch.arange(n_samples, device=device, dtype=torch.float64)
    gen = torch.Generator(device=device)
    for b in range(len(recipes)):
        arg = (2 * np.pi / sr) * f[b][:, None] * n[None, :] + p[b][:, None]
        y = (a[b][:, None] * torch.sin(arg).float()).sum(0)
        gen.manual_seed(recipes[b][3])
        y += torch.randn(n_samples, device=device, generator=gen) * 0.003
        out[b] = y * (0.9 / y.abs().max())
    labels = {k: np.stack([r[4][k] for r in recipes]) for k in recipes[0][4]}
    return out, labels
