"""Key-signature table and the MIREX key score, batched.

Same results as the reference's per-sample Python loop (``PitchClassNet.mirex_score``,
models.py:1065-1116, table utils/key_signatures.py:19-42) but evaluated for the whole batch
at once on whatever device the predictions live on (no per-sample table upload, models.py:1077).
"""
from __future__ import annotations

import torch


def _build_table() -> torch.Tensor:
    # circle of fifths Cb .. C# (15 rows) + six enharmonic "theoretical" keys
    rows = []
    for i in range(15):
        tonic = (7 * (i - 7)) % 12
        rows.append([1.0 if ((pc - tonic) % 12) in (0, 2, 4, 5, 7, 9, 11) else 0.0 for pc in range(12)])
    rows += [rows[j] for j in (9, 11, 10, 4, 3, 5)]
    return torch.tensor(rows, dtype=torch.float32)


KEY_SIGNATURE_MAP = _build_table()      # (21, 12)


_tables = {}


def _device_table(dev, dtype):
    """The table on `dev`, uploaded once (the reference re-uploads it per sample, models.py:1077; a host-to-device copy per call
    also stalls the launch queue and cannot be captured into a graph)."""
    key = (str(dev), dtype)
    if key not in _tables:
        _tables[key] = KEY_SIGNATURE_MAP.to(device=dev, dtype=dtype)
    return _tables[key]


def mirex_score(key_labels, key_preds, tonic_labels, tonic_preds, key_signature_id):
    """-> (mirex, correct, fifths, relative, parallel, other, accuracy), float32 scalars.

    Category logic and its quirks follow models.py:1084-1114 exactly: the predicted key is the
    first-maximum cosine match over the 21-row table, ``diff`` compares that row index with the
    24-way label index, and the if-chain gives 'fifths' precedence.
    """
    dev = key_preds.device
    table = _device_table(dev, key_preds.dtype)
    eps = 1e-8
    pn = key_preds.norm(dim=1, keepdim=True).clamp_min(eps)
    tn = table.norm(dim=1, keepdim=True).clamp_min(eps)
    sims = (key_preds @ table.T) / (pn * tn.T)                               # (B, 21)
    # torch.argmax returns the first maximum on CPU and CUDA alike for exact ties only when
    # computed on identical values; duplicates rows (0/12, 1/13, ...) give bit-identical sims.
    pred_id = torch.argmax(sims, dim=1)
    first = torch.full_like(pred_id, table.shape[0])
    is_max = sims == sims.gather(1, pred_id[:, None])
    idx = torch.arange(table.shape[0], device=dev)[None, :].expand_as(sims)
    pred_id = torch.where(is_max, idx, first[:, None].expand_as(sims)).min(dim=1).values
    key_pred = table[pred_id]
    label_id = torch.argmax(key_signature_id, dim=1)
    correct_keys = (key_pred == key_labels.to(key_pred.dtype)).sum(dim=1)
    full = correct_keys == 12
    diff = (pred_id - label_id).abs()
    tonic_ok = torch.argmax(tonic_labels, dim=1) == torch.argmax(tonic_preds, dim=1)
    fifths = (diff == 1) & ~(tonic_ok & full)
    correct = tonic_ok & full & ~fifths
    relative = full & ~tonic_ok & ~fifths
    parallel = tonic_ok & ~full & ~fifths
    other = ~(fifths | correct | relative | parallel)
    n = float(key_preds.shape[0])
    f = lambda m: (m.sum().float() / n).float()
    mirex = (1.0 * correct.sum() + 0.5 * fifths.sum() + 0.3 * relative.sum() + 0.2 * parallel.sum()).float() / n
    return mirex.float(), f(correct), f(fifths), f(relative), f(parallel), f(other), f(full)
