"""PARITY (GPU): ake_general_step_f32 -- general_step's loss, its gradient with respect to the network outputs and the nine metrics in
one launch (models.py:826-905, 1065-1116) -- against the reference-generated fixture (tests/golden/mirex_loss_cases.npz: the
reference's own loss / step metrics / MIREX categories), the numpy oracles, and float64 torch autograd of the written-out loss."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import ake_amd
from ake_amd import _lib
from oracle import loss_oracle, mirex_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def run(key, tonic, genre, key_labels, tonic_labels, genre_labels, sig, weights=(1.0, 1.0, 0.1), use_cos=False, grads=True):
    """-> (scalars[10] numpy, d_key, d_tonic, d_genre numpy or None) straight through the C ABI."""
    L = _lib.lib()
    dev = lambda t: None if t is None else (t if torch.is_tensor(t) else torch.from_numpy(np.ascontiguousarray(t))).to(DEV).contiguous()
    key, tonic, genre, key_labels = (None if t is None else dev(t).float() for t in (key, tonic, genre, key_labels))
    tl, gl, sl = dev(tonic_labels), dev(genre_labels), dev(sig)
    B = key.shape[0]
    scal = torch.full((10,), float("nan"), device=DEV)
    dk, dt = (torch.full((B, 12), float("nan"), device=DEV) for _ in range(2)) if grads else (None, None)
    dg = torch.full((B, 11), float("nan"), device=DEV) if grads and genre is not None else None
    p = lambda t: t.data_ptr() if t is not None else None
    i64 = lambda t: int(t is not None and t.dtype == torch.int64)
    _lib.check(L.ake_general_step_f32(p(key), p(tonic), p(genre), p(key_labels), p(tl), i64(tl), p(gl), i64(gl), p(sl), i64(sl), B,
                                      weights[0], weights[1], weights[2], int(use_cos), p(scal), p(dk), p(dt), p(dg), None), "ake_general_step_f32")
    torch.cuda.synchronize()
    n = lambda t: None if t is None else t.cpu().numpy()
    return n(scal), n(dk), n(dt), n(dg)


def test_reference_step_fixture(gold_default, gold_mirex):
    """The reference's own general_step on its own outputs: loss and the nine metrics."""
    g = gold_mirex
    scal, *_ = run(gold_default["key"], gold_default["tonic"], gold_default["genre"], g["loss_key_labels"], g["loss_tonic_labels"],
                   g["loss_genre_labels"], g["loss_sig"])
    assert abs(scal[0] - float(g["loss"])) < 2e-6 * abs(float(g["loss"]))
    assert np.allclose(scal[1:], g["step_metrics"], atol=1e-7), (scal[1:], g["step_metrics"])


def test_reference_mirex_fixture(gold_mirex):
    """96 labelled predictions of the reference's fixture (every category occurs): batch scores and each sample on its own."""
    g = gold_mirex
    order = [2, 3, 4, 5, 6, 7, 1]          # fixture: mirex, correct, fifths, relative, parallel, other, accuracy
    kp, tp = g["key_preds"].astype(np.float32), g["tonic_preds"].astype(np.float32)
    scal, *_ = run(kp, tp, None, g["key_labels"], g["tonic_labels"], None, g["key_signature_id"], grads=False)
    assert np.allclose(scal[order], g["mirex"], atol=1e-7)
    for i in range(len(kp)):
        s = slice(i, i + 1)
        scal, *_ = run(kp[s], tp[s], None, g["key_labels"][s], g["tonic_labels"][s], None, g["key_signature_id"][s], grads=False)
        assert np.array_equal(scal[order], g["mirex_per_sample"][i]), i


def torch_step(key, tonic, genre, key_labels, tonic_labels, genre_labels, weights, use_cos):
    """models.py:826-896 written out with float64 torch ops (autograd gives the output gradients)."""
    key, tonic = key.double().requires_grad_(True), tonic.double().requires_grad_(True)
    genre = genre.double().requires_grad_(True) if genre is not None else None
    loss = weights[0] * F.binary_cross_entropy(key, key_labels.double()) + weights[1] * F.cross_entropy(tonic, tonic_labels.long().argmax(1))
    if genre is not None:
        gl = genre_labels.long()
        mask = gl.sum(1) == 1
        if mask.sum() != 0:
            loss = loss + weights[2] * F.cross_entropy(genre[mask], gl.argmax(1)[mask])
    if use_cos:
        loss = loss + (1 - F.cosine_similarity(key, key_labels.double(), dim=1).sum() / key.shape[0])
    loss.backward()
    return float(loss.detach()), key.grad, tonic.grad, (genre.grad if genre is not None and genre.grad is not None else None)


@pytest.mark.parametrize("batch,with_genre,use_cos,i64", [(37, True, False, False), (8, True, True, True), (300, False, False, True), (1, True, False, False),
                                                         (5, True, False, True)])
def test_random_batches_against_oracles_and_autograd(batch, with_genre, use_cos, i64):
    g = torch.Generator().manual_seed(batch)
    key = torch.rand((batch, 12), generator=g) * 0.98 + 0.01
    tonic = torch.randn((batch, 12), generator=g) * 2
    genre = torch.randn((batch, 11), generator=g) * 2 if with_genre else None
    kid = torch.randint(0, 24, (batch,), generator=g)
    rows = ake_amd.KEY_SIGNATURE_MAP
    key_labels = rows[torch.randint(0, 21, (batch,), generator=g)].clone()
    key[::3] = key_labels[::3] * 0.9 + 0.05                       # a third of the rows predict their label's scale (correct / relative / fifths occur)
    tonic_labels = F.one_hot(torch.randint(0, 12, (batch,), generator=g), 12)
    tonic[::2] += 6 * tonic_labels[::2]
    sig = F.one_hot(kid, 24)
    genre_labels = F.one_hot(torch.randint(0, 11, (batch,), generator=g), 11) if with_genre else None
    if with_genre:
        genre_labels[1::4] = 0                                    # rows without a genre label
        if batch == 5:
            genre_labels[:] = 0                                   # no labelled row at all: the genre term is dropped (models.py:892)
    cast = (lambda t: t.long()) if i64 else (lambda t: t.float())
    weights = (1.0, 0.7, 0.1)
    scal, dk, dt, dg = run(key, tonic, genre, key_labels, cast(tonic_labels), cast(genre_labels) if with_genre else None, cast(sig), weights, use_cos)
    # loss: numpy oracle (pinned on the reference's loss) and autograd
    ref = loss_oracle.general_step_loss(key.numpy(), tonic.numpy(), genre.numpy() if with_genre else None, key_labels.numpy(), tonic_labels.numpy(),
                                        genre_labels.numpy() if with_genre else None, *weights, use_cos=use_cos)
    lt, gk, gt, gg = torch_step(key, tonic, genre, key_labels, tonic_labels, genre_labels, weights, use_cos)
    assert abs(ref - lt) < 1e-12 and abs(scal[0] - ref) < 2e-6 * max(1.0, abs(ref))
    assert np.abs(dk - gk.numpy()).max() < 2e-6 * max(1e-3, float(gk.abs().max()))
    assert np.abs(dt - gt.numpy()).max() < 2e-6 * max(1e-3, float(gt.abs().max()))
    if with_genre:
        want = gg.numpy() if gg is not None else np.zeros((batch, 11))
        assert np.abs(dg - want).max() < 2e-6 * max(1e-3, float(np.abs(want).max()))
    # metrics: the MIREX oracle (pinned on the reference's per-sample loop) on the same float32 predictions
    m = mirex_oracle.mirex_score(key_labels.numpy(), key.numpy(), tonic_labels.numpy(), tonic.numpy(), sig.numpy())
    assert np.allclose(scal[[2, 3, 4, 5, 6, 7, 1]], np.array(m, np.float32), atol=1e-6)
    assert abs(scal[8] - float((tonic.argmax(1) == tonic_labels.argmax(1)).float().mean())) < 1e-6
    if with_genre:
        mask = genre_labels.sum(1) == 1
        acc_g = float(((genre.argmax(1) == genre_labels.argmax(1)) & mask).sum()) / max(1, int(mask.sum()))
        assert abs(scal[9] - acc_g) < 1e-6
    else:
        assert scal[9] == 0


@pytest.mark.parametrize("mel_dtype", [torch.float32, torch.float64])
def test_general_step_takes_the_device_path_and_matches_the_torch_ops(gold_default, mel_dtype):
    """PitchClassNet.general_step on the GPU goes through the kernel; the same step with the torch ops (a subclass overriding
    mirex_score keeps them) returns the same ten values and the same parameter gradients."""
    import json
    from argparse import Namespace
    from conftest import golden_state_dict

    class TorchOps(ake_amd.PitchClassNet):
        def mirex_score(self, *a):
            return super().mirex_score(*a)

    opt = Namespace(**json.loads(str(gold_default["opt"])))
    g = torch.Generator().manual_seed(3)
    B, T = 6, 40
    key_id = torch.randint(0, 24, (B,), generator=g)
    batch = {"mel": (torch.rand((B, 1, 288, T), generator=g) * 2.5).to(DEV).to(mel_dtype), "seq_length": torch.tensor([T, T - 3, T, T - 9, T, T]),
             "key_labels": ake_amd.KEY_SIGNATURE_MAP[key_id % 21], "tonic_labels": F.one_hot(key_id % 12, 12).float(),
             "key_signature_id": F.one_hot(key_id, 24).float(), "genre": F.one_hot(torch.randint(0, 11, (B,), generator=g), 11)}
    batch["genre"][2] = 0
    res = []
    for cls in (ake_amd.PitchClassNet, TorchOps):
        net = cls(288, 12, 2, 7, opt)
        net.load_state_dict(golden_state_dict(gold_default), strict=True)
        net = net.to(DEV).train()
        vals = net.general_step(batch, 0, "train")
        assert (vals[0].grad_fn is not None) and (("FusedGeneralStep" in type(vals[0].grad_fn.next_functions[0][0]).__name__) == (cls is ake_amd.PitchClassNet))
        vals[0].backward()
        res.append(([float(v.detach()) for v in vals], torch.cat([p.grad.reshape(-1) for p in net.parameters()]).cpu()))
    (va, ga), (vb, gb) = res
    assert np.allclose(va, vb, rtol=2e-6, atol=1e-7), (va, vb)
    assert float((ga - gb).abs().max()) < 1e-5 * float(gb.abs().max())
