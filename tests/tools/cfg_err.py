#!/usr/bin/env python3
"""Error of the non-default configurations against the float64 oracle, fused and unfused (debug): python3 tests/tools/cfg_err.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from argparse import Namespace
import torch
import ake_amd
from oracle import pcnet_oracle

def rel(a, b):
    return float((a.double().cpu() - b).abs().max() / b.abs().max())

for cfg in [dict(num_layers=1), dict(num_layers=3), dict(head_layers=1), dict(head_layers=3), dict(conv_layers=2, n_filters=2), dict(n_filters=3),
            dict(max_pool=True), dict(time_pool_size=4), dict()]:
    name = str(cfg)
    opt = Namespace(conv_layers=3, n_filters=4, head_layers=2, time_pool_size=2, genre=True, max_pool=False, frames=5)
    num_layers = cfg.pop("num_layers", 2)
    for k, v in cfg.items():
        setattr(opt, k, v)
    torch.manual_seed(11)
    net = ake_amd.PitchClassNet(288, 12, num_layers, 7, opt)
    g = torch.Generator().manual_seed(3)
    for _, mod in net.named_modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.2)
            mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
            mod.weight.data.copy_(torch.rand(mod.weight.shape, generator=g) + 0.5)
            mod.bias.data.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
    sd64 = pcnet_oracle.to_dtype(net.state_dict(), torch.float64)
    T = 120 if (num_layers == 3 or opt.time_pool_size == 4 or opt.head_layers == 3) else 52
    x = torch.rand((3, 1, 288, T), generator=g) * 2.5
    seq = torch.tensor([T, T - 9, T - 20])
    ref = pcnet_oracle.pcnet_forward(sd64, x.double(), seq, head_layers=opt.head_layers, time_pool_size=opt.time_pool_size, max_pool=opt.max_pool)
    net = net.cuda().eval()
    res = []
    for keep in (False, True):
        was = net.keep_taps(keep)
        got = net(x.cuda(), seq.cuda())
        net.keep_taps(was)
        res.append([rel(a, b) for a, b in zip(got, ref)])
    os.environ.pop("X", None)
    print(f"{name:40s} fused {['%.1e' % e for e in res[0]]}   unfused {['%.1e' % e for e in res[1]]}")
