#!/usr/bin/env python3
"""Batch statistics of every BatchNorm for k shuffled copies of a 32-clip batch against those of the batch itself (both on the device):
they are the same numbers up to the rounding of other partial sums."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import torch
from conftest import load_golden
import test_gpu_train_scale as t

gold = load_golden("pcnet_default.npz")
def stats(copies):
    net, _ = t.fresh_net(gold)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.zero_(); m.running_var.zero_()
    x, seq, labels = t.big_case(32, 76, 0)
    if copies > 1:
        x, seq, labels = t.replicate(x, seq, labels, copies, 5)
    with torch.no_grad():
        net(x.to(t.DEV), seq.to(t.DEV))
    n = x.shape[0]
    return {k: v.detach().cpu().double() for k, v in net.state_dict().items() if "running" in k}
base = stats(1)
for copies in (2, 8):
    s = stats(copies)
    print("copies", copies)
    for k in base:
        a, b = s[k], base[k]
        if k.endswith("running_var"):
            # unbiased variance: n/(n-1) differs between the batch sizes; compare the biased ones
            pass
        rel = ((a - b).abs() / b.abs().clamp_min(1e-12)).max()
        print(f"   {float(rel):9.2e}  {k}")
