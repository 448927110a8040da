"""CPU restatement of the training loss in ``general_step``.  TEST INFRASTRUCTURE ONLY.

Follows models.py:855-896 for the global (``opt.local`` False) branch, in numpy
float64 with the formulas written out (no torch loss modules), so it is an
independent check of both the reference and the device loss kernel.
"""
from __future__ import annotations

import numpy as np


def _log_softmax(z):
    z = z - z.max(axis=1, keepdims=True)
    return z - np.log(np.exp(z).sum(axis=1, keepdims=True))


def bce(p, y):
    """nn.BCELoss() mean reduction; log clamped at -100 as torch does (models.py:855,878)."""
    lp = np.maximum(np.log(p), -100.0)
    l1p = np.maximum(np.log1p(-p), -100.0)
    return float(-(y * lp + (1.0 - y) * l1p).mean())


def cross_entropy(logits, target_idx):
    """nn.CrossEntropyLoss() mean reduction over rows (models.py:856-857,879,883)."""
    ls = _log_softmax(logits)
    return float(-ls[np.arange(len(target_idx)), target_idx].mean())


def general_step_loss(key_out, tonic_out, genre_out, key_labels, tonic_labels, genre_labels,
                      key_weight=1.0, tonic_weight=1.0, genre_weight=0.1, use_cos=False):
    """loss of models.py:878-896.

    ``genre_out``/``genre_labels`` may be None (``opt.genre`` False).  Genre rows
    whose one-hot label does not sum to 1 are masked out (:839, :881-883); if no
    row survives the genre term is dropped (:892-893).
    """
    key_out = np.asarray(key_out, np.float64)
    tonic_out = np.asarray(tonic_out, np.float64)
    key_labels = np.asarray(key_labels, np.float64)
    tonic_idx = np.argmax(np.asarray(tonic_labels), axis=1)                # :832
    loss = key_weight * bce(key_out, key_labels) + tonic_weight * cross_entropy(tonic_out, tonic_idx)  # :889
    if genre_out is not None:
        gl = np.asarray(genre_labels).astype(np.int64)                     # :826 (.long())
        mask = gl.sum(axis=1) == 1                                         # :839
        if mask.sum() != 0:                                                # :892
            g = np.asarray(genre_out, np.float64)[mask]
            loss += genre_weight * cross_entropy(g, np.argmax(gl, axis=1)[mask])   # :881-883, :893
    if use_cos:                                                            # :885-887, :895-896
        num = (key_out * key_labels).sum(1)
        den = np.maximum(np.sqrt((key_out ** 2).sum(1)), 1e-8) * np.maximum(np.sqrt((key_labels ** 2).sum(1)), 1e-8)
        loss += 1.0 - float((num / den).sum()) / key_out.shape[0]
    return loss
