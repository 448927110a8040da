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


F32_BAND = (5e-2, 1e-2)     # (worst tensor, median tensor): what float32 PyTorch on the CPU itself shows against float64 at these sizes, see below


def test_gradients_at_32_clips(gold_default):
    """Six data seeds at 32 x 76 against float64 autograd.

    At this size a float32 run cannot be tight in every tensor on every seed: with 2.4e7 LeakyReLU / max decisions per step a handful sit
    within 1e-6 of the kink, take the other branch than the float64 forward and move every gradient upstream of them by ~1e-3.  Stock
    float32 PyTorch on the CPU, same batches, against the same float64 gradients: worst tensor 4.6e-3 .. 3.0e-2, MEDIAN tensor 1.0e-3 ..
    3.6e-3 on every one of seeds 0-3 (measured; 256 clips: 5.2e-3 / 2.1e-3) -- the device is usually far inside that (median 7e-6 on seeds
    without a flip near the top of the net).  So the statement that separates a kernel defect from such flips:
      * every seed: inside the float32 band (a defect that scales with the data, e.g. a wrong tile, breaks it);
      * every TENSOR is tight (3e-5 of its max) on at least one seed: a flip spoils the tensors upstream of it on ITS seed only, a
        systematic error of a kernel -- a tile dropped by the persistent loop, a partial lost in the ordered reduction, low bits of dz
        lost -- spoils its tensors on every seed;
      * at least two seeds have a tight median."""
    net, sd32 = fresh_net(gold_default)
    best, tight_median = {}, 0
    for seed in range(6):
        x, seq, labels = big_case(32, 76, seed)
        loss_ref, ref = oracle_grads(sd32, x, seq, labels)
        loss, got = device_grads(net, x, seq, labels)
        assert abs(loss - loss_ref) < 2e-5 * max(1.0, abs(loss_ref)), (seed, loss, loss_ref)
        rows = error_rows(got, ref)
        report(f"32 clips, seed {seed}", rows)
        assert rows[0][0] < F32_BAND[0] and rows[len(rows) // 2][0] < F32_BAND[1], (seed, rows[:4])
        # no tensor is "effectively unchecked" (ADVICE r2): relative to its OWN largest entry every gradient tensor is right to 10 %
        assert max(r[3] for r in rows if r[1] not in ILL_CONDITIONED) < 1e-1, (seed, sorted(rows, key=lambda r: -r[3])[:4])
        tight_median += rows[len(rows) // 2][0] < 3e-5
        for e, n, _, _ in rows:
            best[n] = min(best.get(n, 1.0), e)
    never_tight = sorted(((e, n) for n, e in best.items() if e > ILL_CONDITIONED.get(n, 3e-5)), reverse=True)
    print("\nbest error of every tensor over the seeds: worst", max(best.values()))
    assert not never_tight, never_tight[:6]
    assert tight_median >= 2, tight_median


def replicate(x, seq, labels, copies, seed):
    """`copies` copies of every clip, in a random order: the batch statistics of every BatchNorm, the per-clip activations and the (mean)
    losses are those of the small batch, so the loss is the same FUNCTION of the weights and the gradients are equal."""
    perm = torch.randperm(x.shape[0] * copies, generator=torch.Generator().manual_seed(seed))
    rep = lambda t: t.repeat((copies,) + (1,) * (t.dim() - 1))[perm]
    return rep(x), rep(seq), tuple(rep(t) for t in labels)


def test_bench_shard_gradients_equal_those_of_the_replicated_32_clips(gold_default):
    """The benchmarked shard without any float64 noise floor: 256 clips x 76 frames = 8 shuffled copies of the 32-clip batch above must
    give the 32-clip gradients (same loss function of the weights; LeakyReLU / max decisions are taken on the same values, to the 1e-7 that
    the sums over 8x as many elements round differently).  What differs is everything the large launch does differently: 7 424 row tiles
    on 256 persistent workgroups (29 per workgroup: prefetch, buffer swap, pending stores, statistics flush), 512 weight-gradient partials
    per convolution, BatchNorm sums over 5.6 M values per channel, dz of 1e-7 (the f16 hi / lo data-gradient operands)."""
    net, _ = fresh_net(gold_default)
    x, seq, labels = big_case(32, 76, 0)
    loss32, g32 = device_grads(net, x, seq, labels)
    x8, seq8, labels8 = replicate(x, seq, labels, 8, 5)
    loss256, g256 = device_grads(net, x8, seq8, labels8)
    assert abs(loss256 - loss32) < 2e-6 * max(1.0, abs(loss32)), (loss256, loss32)
    rows = error_rows(g256, g32)
    report("256 = 8 x 32 clips against 32 clips (device both)", rows)
    assert rows[0][0] < 2e-3 and rows[len(rows) // 2][0] < 2e-5, rows[:5]


def test_gradients_at_the_bench_shard(gold_default):
    """256 clips x 76 frames of DISTINCT data -- the per-rank batch of `bench.py --train` and of BASELINE configs[3] -- against float64
    autograd of the whole batch (about a minute on the host).  Inside the float32 band (float32 PyTorch on the CPU, this batch: worst
    5.2e-3, median 2.1e-3 against the same float64 gradients); the tensors nothing can flip above -- the heads' last convolutions -- tight."""
    net, sd32 = fresh_net(gold_default)
    x, seq, labels = big_case(256, 76, 7)
    loss_ref, ref = oracle_grads(sd32, x, seq, labels)
    loss, got = device_grads(net, x, seq, labels)
    assert abs(loss - loss_ref) < 2e-5 * max(1.0, abs(loss_ref)), (loss, loss_ref)
    rows = error_rows(got, ref)
    report("256 clips", rows)
    assert rows[0][0] < F32_BAND[0] and rows[len(rows) // 2][0] < F32_BAND[1], rows[:4]
    assert max(r[3] for r in rows if r[1] not in ILL_CONDITIONED) < 1e-1, sorted(rows, key=lambda r: -r[3])[:4]
    by_name = {n: e for e, n, _, _ in rows}
    for n in ("key_classifier.3.conv2d.weight", "tonic_classifier.3.conv2d.weight", "genre_classifier.3.weight"):
        assert by_name[n] < 3e-5, (n, by_name[n])


def test_gradients_scale_with_the_loss(gold_default):
    """grad(2^-10 loss) == 2^-10 grad(loss): a power-of-two factor is exact in every float operation of the backward pass, so the two
    differ only where an operand's absolute size matters -- the 2^-40 resolution of the fixed-point sums (1e-12) and, before the fix, the
    f16 hi / lo planes of dz.  At 32 x 76 the unscaled dz are already ~1e-5; times 2^-10 they are where 256 clips and genre_weight put them."""
    net, _ = fresh_net(gold_default)
    x, seq, labels = big_case(32, 76, 1)
    _, g1 = device_grads(net, x, seq, labels)
    _, g2 = device_grads(net, x, seq, labels, scale=2.0 ** -10)
    floor = 1e-4 * max(float(g.abs().max()) for g in g1.values())      # (the tonic head's last bias: sum of softmax - onehot, zero to rounding)
    errs = sorted(((float((g2[n] * 1024.0 - g1[n]).abs().max()) / max(float(g1[n].abs().max()), floor), n) for n in g1), reverse=True)
    print(f"\nloss-scale homogeneity: worst {errs[0][0]:.2e} ({errs[0][1]}), median {errs[len(errs) // 2][0]:.2e}")
    assert errs[0][0] < 2e-4, errs[:5]


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
