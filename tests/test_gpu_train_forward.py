"""PARITY (GPU): training-mode forward (BatchNorm with batch statistics) against the oracle and the reference fixture.

The oracle's train-mode path is itself pinned against the reference (tests/golden/pcnet_guard360.npz was produced by
the reference net in train() mode, exactly as equivariance_test.py:178-203 runs it).
"""
import json
from argparse import Namespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import ake_amd
from conftest import golden_state_dict, rel_err
from oracle import mirex_oracle, pcnet_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 2e-4     # asserted; budget 1e-3.  (fp32 batch statistics over up to 5.6 M values per channel)


def make_net(gold):
    opt = Namespace(**json.loads(str(gold["opt"])))
    net = ake_amd.PitchClassNet(opt.octaves * 36, 12, opt.num_layers, opt.kernel_size, opt)
    net.load_state_dict(golden_state_dict(gold), strict=True)
    return net.to(DEV), opt


def test_train_mode_forward_against_oracle(gold_default):
    net, _ = make_net(gold_default)
    net.train()
    sd = golden_state_dict(gold_default, torch.float64)
    x = torch.from_numpy(gold_default["x"])
    seq = torch.from_numpy(gold_default["seq_length"])
    ref = pcnet_oracle.pcnet_forward(sd, x.double(), seq, training=True)
    got = net(x.to(DEV), seq.to(DEV))
    for a, b, name in zip(got, ref, ("key", "tonic", "genre")):
        assert rel_err(a.cpu(), b) < TOL, name
    # and it differs from the eval-mode result (the statistics really are the batch's)
    assert rel_err(got[1].cpu(), gold_default["tonic"]) > 1e-2


def test_running_statistics_update(gold_default):
    net, _ = make_net(gold_default)
    net.train()
    x = torch.from_numpy(gold_default["x"])
    bn = net.model[0].pool_semi_b
    rm0, rv0, nb0 = bn.running_mean.clone().cpu(), bn.running_var.clone().cpu(), int(bn.num_batches_tracked)
    net(x.to(DEV), None)
    conv = net.model[0].pool_semi
    y = F.conv2d(F.pad(x.double(), (1, 1, 0, 0), mode="circular"), conv.weight.detach().cpu().double(), conv.bias.detach().cpu().double(), stride=(3, 1))
    mean, var_u = y.mean(dim=(0, 2, 3)), y.var(dim=(0, 2, 3), unbiased=True)
    assert torch.allclose(bn.running_mean.cpu().double(), 0.9 * rm0.double() + 0.1 * mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var.cpu().double(), 0.9 * rv0.double() + 0.1 * var_u, rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == nb0 + 1
    # every BatchNorm of the net was visited exactly once
    assert all(int(m.num_batches_tracked) == 1 for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d))
    # eval mode now uses the updated running statistics
    net.eval()
    sd = pcnet_oracle.to_dtype({k: v.cpu() for k, v in net.state_dict().items()}, torch.float64)
    ref = pcnet_oracle.pcnet_forward(sd, x.double(), None)
    got = net(x.to(DEV), None)
    for a, b in zip(got, ref):
        assert rel_err(a.cpu(), b) < TOL


def test_equivariance_script_semantics_train_mode(gold_guard):
    """equivariance_test.py:172-205 exactly as written: 360-bin net left in train() mode, B = 1, 25 shifted inputs."""
    net, opt = make_net(gold_guard)
    net.train()
    mel = gold_guard["mel"].astype(np.float64)
    mel_g = np.concatenate([np.zeros((36, 40)), mel, np.zeros((36, 40))])
    seq = torch.tensor(40).reshape(1, 1)
    rows_k, rows_t = [], []
    for i in range(0, 13):
        k, t = net(torch.from_numpy(mirex_oracle.mel_shifting_up(mel_g, i)).reshape(1, 1, 360, 40).to(DEV), seq)
        rows_k.insert(0, k[0].detach().cpu().numpy()); rows_t.insert(0, t[0].detach().cpu().numpy())
    for i in range(1, 13):
        k, t = net(torch.from_numpy(mirex_oracle.mel_shifting_down(mel_g, i)).reshape(1, 1, 360, 40).to(DEV), seq)
        rows_k.append(k[0].detach().cpu().numpy()); rows_t.append(t[0].detach().cpu().numpy())
    K, T = np.stack(rows_k), np.stack(rows_t)
    assert rel_err(K, gold_guard["key_train"]) < TOL and rel_err(T, gold_guard["tonic_train"]) < TOL
    for s in range(1, 13):          # batch statistics are shift-invariant, so the roll identity holds in train mode too
        assert np.abs(K[12 - s] - np.roll(K[12], s)).max() <= 2e-5
        assert np.abs(T[12 + s] - np.roll(T[12], -s)).max() <= 2e-5


@pytest.mark.parametrize("cfg", [dict(num_layers=1), dict(num_layers=3), dict(head_layers=3), dict(n_filters=3), dict(resblock=True),
                                 dict(resblock=True, num_layers=3, n_filters=2, conv_layers=2), dict(resblock=True, num_layers=1)])
def test_train_mode_other_configurations(cfg):
    opt = Namespace(conv_layers=3, n_filters=4, head_layers=2, time_pool_size=2, genre=True, max_pool=False, frames=5)
    num_layers = cfg.pop("num_layers", 2)
    for k, v in cfg.items():
        setattr(opt, k, v)
    torch.manual_seed(5)
    net = ake_amd.PitchClassNet(288, 12, num_layers, 7, opt)
    sd64 = pcnet_oracle.to_dtype(net.state_dict(), torch.float64)
    T = 120 if (num_layers == 3 or opt.head_layers == 3) else 52
    g = torch.Generator().manual_seed(2)
    x = torch.rand((3, 1, 288, T), generator=g) * 2.5
    seq = torch.tensor([T, T - 9, T - 20])
    ref = pcnet_oracle.pcnet_forward(sd64, x.double(), seq, head_layers=opt.head_layers, training=True)
    got = net.to(DEV).train()(x.to(DEV), seq.to(DEV))
    for a, b in zip(got, ref):
        assert rel_err(a.cpu(), b) < 5e-4
