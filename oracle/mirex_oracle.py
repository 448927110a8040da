"""CPU restatement of the MIREX key score, the key-signature table and the
semitone-shift helpers.  TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Plain numpy / pure-Python loops; follows the reference line by line so that
its quirks are kept (they are part of the metric as published).
"""
from __future__ import annotations

import numpy as np

_MAJOR_STEPS = (0, 2, 4, 5, 7, 9, 11)


def key_signature_map() -> np.ndarray:
    """The 21x12 table of ``utils/key_signatures.py:19-42``.

    Rows 0..14 walk the circle of fifths from Cb major (7 flats) to C# major
    (7 sharps): row i is the major scale on tonic ``7*(i-7) mod 12`` (equally its
    relative minor).  Rows 15..20 are the "theoretical keys", enharmonic copies
    of rows 9, 11, 10, 4, 3, 5 (``:36-41``).
    """
    rows = []
    for i in range(15):
        tonic = (7 * (i - 7)) % 12
        row = np.zeros(12, dtype=np.float32)
        for s in _MAJOR_STEPS:
            row[(tonic + s) % 12] = 1.0
        rows.append(row)
    for src in (9, 11, 10, 4, 3, 5):
        rows.append(rows[src].copy())
    return np.stack(rows, 0)


def _cos(a, b, axis, eps=1e-8):
    """torch.nn.CosineSimilarity semantics: x.y / (max(|x|,eps) * max(|y|,eps))."""
    na = np.maximum(np.sqrt((a * a).sum(axis)), eps)
    nb = np.maximum(np.sqrt((b * b).sum(axis)), eps)
    return (a * b).sum(axis) / (na * nb)


def mirex_score(key_labels, key_preds, tonic_labels, tonic_preds, key_signature_id):
    """``PitchClassNet.mirex_score``, models.py:1065-1116.

    Returns ``(mirex, correct, fifths, relative, parallel, other, accuracy)`` as
    float32 scalars (the reference wraps each in ``torch.tensor(..).float()``).
    Kept quirks: ``diff`` compares an index into the 21-row circle-of-fifths
    table with an index into the 24-way chromatic label (:1084-1095); the
    first-max-wins argmax over duplicate rows; the if-chain order (:1100-1113).
    """
    table = key_signature_map().astype(np.float64)
    key_labels = np.asarray(key_labels, dtype=np.float64)
    key_preds = np.asarray(key_preds, dtype=np.float64)
    tonic_labels = np.asarray(tonic_labels)
    tonic_preds = np.asarray(tonic_preds)
    key_signature_id = np.asarray(key_signature_id)
    accuracy = samples = 0
    correct = fifths = parallel = relative = other = 0
    for i in range(len(key_labels)):                                   # :1071
        category = 0
        sims = _cos(key_preds[i][None, :], table, axis=1)              # :1083-1084
        pred_key_id = int(np.argmax(sims))
        key_pred = table[pred_key_id]                                  # :1085
        key_sig_label_id = int(np.argmax(key_signature_id[i]))         # :1086
        correct_keys = int((key_pred == key_labels[i]).sum())          # :1090
        accuracy += 1 if correct_keys == 12 else 0                     # :1092
        samples += 1
        diff = abs(pred_key_id - key_sig_label_id)                     # :1095
        correct_tonic = 1 if int(np.argmax(tonic_labels[i])) == int(np.argmax(tonic_preds[i])) else 0  # :1096
        if diff == 1 and not (correct_tonic == 1 and correct_keys == 12):   # :1100
            fifths += 1
            category = 1
        if correct_tonic == 1 and correct_keys == 12 and category == 0:     # :1103
            correct += 1
            category = 1
        if correct_keys == 12 and correct_tonic == 0 and category == 0:     # :1106
            relative += 1
            category = 1
        if correct_tonic == 1 and correct_keys != 12 and category == 0:     # :1109
            parallel += 1
            category = 1
        if category == 0:                                                   # :1112
            other += 1
    mirex = 1 * correct + 0.5 * fifths + 0.3 * relative + 0.2 * parallel     # :1114
    f = np.float32
    return (f(mirex / samples), f(correct / samples), f(fifths / samples), f(relative / samples),
            f(parallel / samples), f(other / samples), f(accuracy / samples))


def mel_shifting_up(mel: np.ndarray, semitones: int) -> np.ndarray:
    """equivariance_test.py:122-133: move content up by 3*semitones bins, zero-fill below."""
    steps = 3 * semitones
    out = np.zeros_like(mel)
    if steps == 0:
        return mel.copy()
    out[steps:] = mel[:-steps]
    return out


def mel_shifting_down(mel: np.ndarray, semitones: int) -> np.ndarray:
    """equivariance_test.py:135-146: move content down by 3*semitones bins, zero-fill above."""
    steps = 3 * semitones
    out = np.zeros_like(mel)
    if steps == 0:
        return mel.copy()
    out[:-steps] = mel[steps:]
    return out
