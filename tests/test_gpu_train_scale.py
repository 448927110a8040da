"""PARITY (GPU) at the sizes that are BENCHMARKED: the training step of `bench.py --train` / BASELINE configs[3]
runs 256 clips x 76 frames per rank -- 7 424 row tiles on 256 persistent workgroups, hundreds of weight-gradient partials, BatchNorm sums
over 5.6 M values per channel -- while every other gradient test uses 1..6 clips (one tile per workgroup).  Here:

  * gradients of the default net at 32 clips x 76 frames (29 x 32 = 928 tiles: 3-4 tiles per workgroup of conv_p2p_f16x3_kernel, its
    double-buffer swap / pending-store hand-over / statistics flush; 32..64 partials in wgrad_partial_reduce_kernel) against float64
    autograd through the pinned oracle -- the kink-robust statement of tests/test_gpu_backward.py;
  * the same at 256 clips (the benchmark's per-rank shard) -- one seed, float64 autograd of the whole batch on the host (about a minute);
  * the train-mode forward at 256 clips: outputs AND every BatchNorm's batch mean / variance against the float64 oracle forward;
  * gradients are homogeneous in the loss scale: a loss times 2^-10 must give the same gradients times 2^-10 (the f16 hi / lo
    data-gradient operands once lost their low bits for |dz| below f16's normal range).
"""
import json
from argparse import Namespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import ake_amd
from conftest import golden_state_dict, rel_err
from oracle import pcnet_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ILL_CONDITIONED = {"model.0.pool_semi_b.weight": 2e-3}     # see tests/test_gpu_backward.py


def big_case(batch, frames, seed):
    g = torch.Generator().manual_seed(1000 + seed)
    x = torch.rand((batch, 1, 288, frames), generator=g) * 2.5
    seq = frames - torch.randint(0, 20, (batch,), generator=g)
    seq[0] = frames
    key_labels = (torch.rand((batch, 12), generator=g) > 0.5).float()
    tonic_idx = torch.randint(0, 12, (batch,), generator=g)
    genre_idx = torch.randint(0, 11, (batch,), generator=g)
    genre_mask = torch.rand((batch,), generator=g) > 0.25
    genre_mask[0] = True
    return x, seq, (key_labels, tonic_idx, genre_idx, genre_mask)


def loss_fn(key, tonic, genre, key_labels, tonic_idx, genre_idx, genre_mask):
    loss = F.binary_cross_entropy(key, key_labels.to(key.dtype)) + F.cross_entropy(tonic, tonic_idx)     # models.py:878-889
    return loss + 0.1 * F.cross_entropy(genre[genre_mask], genre_idx[genre_mask])                         # models.py:881-893


def oracle_grads(sd32, x, seq, labels):
    sd = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.double() if v.is_floating_point() else v)
          for k, v in sd32.items()}
    out = pcnet_oracle.pcnet_forward(sd, x.double(), seq, training=True)
    loss = loss_fn(out[0], out[1], out[2], *labels)
    loss.backward()
    return float(loss.detach()), {k: v.grad for k, v in sd.items() if torch.is_tensor(v) and v.requires_grad}


def fresh_net(gold_default):
    opt = Namespace(**json.loads(str(gold_default["opt"])))
    net = ake_amd.PitchClassNet(288, 12, 2, 7, opt)
    sd32 = golden_state_dict(gold_default)
    net.load_state_dict(sd32, strict=True)
    return net.to(DEV).train(), sd32


def device_grads(net, x, seq, labels, scale=1.0):
    for p in net.parameters():
        p.grad = None
    out = net(x.to(DEV), seq.to(DEV))
    loss = loss_fn(out[0], out[1], out[2], *(t.to(DEV) for t in labels))
    (loss * scale).backward()
    return float(loss.detach()), {n: p.grad.detach().cpu().double() for n, p in net.named_parameters()}


def error_rows(got, ref):
    """[(max|g - ref| / max(max|ref|, floor), name, max|ref|, unfloored ratio)], worst first; floor = 1e-4 of the step's largest gradient
    (a tensor four orders below it is a cancelling sum whose absolute accuracy the others set, tests/test_gpu_backward.py)."""
    floor = 1e-4 * max(float(ref[n].abs().max()) for n in got)
    rows = []
    for n, g in got.items():
        r = ref[n]
        m = float(r.abs().max())
        if n.endswith(".bias") and m < 1e-9:          # a convolution bias in front of a BatchNorm: exactly zero gradient
            assert float(g.abs().max()) < 1e-6, n
            continue
        d = float((g - r).abs().max())
        rows.append((d / max(m, floor, 1e-7), n, m, d / max(m, 1e-30)))
    rows.sort(reverse=True)
    return rows


def report(tag, rows):
    print(f"\n{tag}: worst {rows[0][0]:.2e} ({rows[0][1]}), median {rows[len(rows) // 2][0]:.2e}, "
          f"worst unfloored {max(r[3] for r in rows):.2e}")


def test_gradients_at_32_clips(gold_default):
    """Four data seeds at 32 x 76.  Every seed: inside the kink-damage bound with a tight median; at least two seeds tight in EVERY tensor
    (a systematic error of a kernel -- a tile dropped in the persistent loop, a partial lost in the reduction -- fails all four)."""
    net, sd32 = fresh_net(gold_default)
    tight = 0
    for seed in range(4):
        x, seq, labels = big_case(32, 76, seed)
        loss_ref, ref = oracle_grads(sd32, x, seq, labels)
        loss, got = device_grads(net, x, seq, labels)
        assert abs(loss - loss_ref) < 2e-5 * max(1.0, abs(loss_ref)), (seed, loss, loss_ref)
        rows = error_rows(got, ref)
        report(f"32 clips, seed {seed}", rows)
        assert rows[0][0] < 5e-2 and rows[len(rows) // 2][0] < 1e-4, (seed, rows[:4])
        # no tensor is "effectively unchecked" (ADVICE r2): even relative to its OWN largest entry every gradient tensor is right to 2 %
        assert max(r[3] for r in rows if r[1] not in ILL_CONDITIONED) < 1e-1, (seed, sorted(rows, key=lambda r: -r[3])[:4])
        tight += all(e <= ILL_CONDITIONED.get(n, 3e-5) for e, n, _, _ in rows)
    assert tight >= 2, tight


def test_gradients_at_the_bench_shard(gold_default):
    """256 clips x 76 frames -- the per-rank batch of `bench.py --train` and of BASELINE configs[3]: 7 424 tiles over 256 workgroups (29 per
    workgroup), 512 partials per weight gradient, 5.6 M values per BatchNorm channel; dz around 1e-7."""
    net, sd32 = fresh_net(gold_default)
    x, seq, labels = big_case(256, 76, 7)
    loss_ref, ref = oracle_grads(sd32, x, seq, labels)
    loss, got = device_grads(net, x, seq, labels)
    assert abs(loss - loss_ref) < 2e-5 * max(1.0, abs(loss_ref)), (loss, loss_ref)
    rows = error_rows(got, ref)
    report("256 clips", rows)
    assert rows[0][0] < 5e-2 and rows[len(rows) // 2][0] < 1e-4, rows[:4]
    assert max(r[3] for r in rows if r[1] not in ILL_CONDITIONED) < 1e-1, sorted(rows, key=lambda r: -r[3])[:4]


def test_gradients_scale_with_the_loss(gold_default):
    """grad(2^-10 loss) == 2^-10 grad(loss): a power-of-two factor is exact in every float operation of the backward pass, so the two
    differ only where an operand's absolute size matters -- the 2^-40 resolution of the fixed-point sums (1e-12) and, before the fix, the
    f16 hi / lo planes of dz.  At 32 x 76 the unscaled dz are already ~1e-5; times 2^-10 they are where 256 clips and genre_weight put them."""
    net, _ = fresh_net(gold_default)
    x, seq, labels = big_case(32, 76, 1)
    _, g1 = device_grads(net, x, seq, labels)
    _, g2 = device_grads(net, x, seq, labels, scale=2.0 ** -10)
    worst = 0.0
    for n in g1:
        m = float(g1[n].abs().max())
        if m < 1e-9:
            continue
        e = float((g2[n] * 1024.0 - g1[n]).abs().max()) / m
        worst = max(worst, e)
        assert e < 2e-4, (n, e, m)
    print(f"\nloss-scale homogeneity: worst {worst:.2e}")


def test_train_forward_and_batch_statistics_at_the_bench_batch(gold_default):
    """Train-mode forward of 256 x 76: the three outputs and EVERY BatchNorm's batch mean / biased variance (recovered from the running
    statistics the step leaves: running = 0.9 old + 0.1 batch) against the float64 oracle forward of the same batch."""
    net, sd32 = fresh_net(gold_default)
    x, seq, _ = big_case(256, 76, 3)
    sd64 = pcnet_oracle.to_dtype(sd32, torch.float64)
    with torch.no_grad(), pcnet_oracle.record_bn_stats() as rows:
        ref = pcnet_oracle.pcnet_forward(sd64, x.double(), seq, training=True)
    with torch.no_grad():
        got = net(x.to(DEV), seq.to(DEV))
    for a, b, name in zip(got, ref, ("key", "tonic", "genre")):
        assert rel_err(a.cpu(), b) < 2e-4, (name, rel_err(a.cpu(), b))
    new = {k: v.detach().cpu().double() for k, v in net.state_dict().items()}
    assert len(rows) == sum(1 for k in sd32 if k.endswith("running_mean"))
    worst_m = worst_v = 0.0
    for prefix, mean, var, count in rows:
        m_dev = (new[prefix + "running_mean"] - 0.9 * sd64[prefix + "running_mean"]) / 0.1
        v_dev = (new[prefix + "running_var"] - 0.9 * sd64[prefix + "running_var"]) / 0.1 * (count - 1) / count
        std = var.sqrt()
        em = float(((m_dev - mean).abs() / (std + 1e-6)).max())          # the mean in units of the channel's standard deviation
        ev = float(((v_dev - var).abs() / (var + 1e-12)).max())
        worst_m, worst_v = max(worst_m, em), max(worst_v, ev)
        # (running = 0.9 old + 0.1 batch is stored in f32: recovering the batch value loses a factor 10 of its 6e-8)
        assert em < 2e-5 and ev < 5e-5, (prefix, em, ev)
        assert int(new[prefix + "num_batches_tracked"]) == int(sd32[prefix + "num_batches_tracked"]) + 1
    print(f"\nbatch statistics at 256 clips: mean {worst_m:.2e} sigma, variance {worst_v:.2e} relative")
