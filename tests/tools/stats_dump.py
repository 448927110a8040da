#!/usr/bin/env python3
"""Dump (or compare with a dump) the running statistics and outputs one train-mode forward of seed-0's 32 clips leaves: bit-level check that the
forward does not depend on the tiling (AKE_USE_DIAG_LIB=1 AKE_TILING_CUS=32).   python3 tests/tools/stats_dump.py <file> [compare]"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import torch
from conftest import load_golden
import test_gpu_train_scale as t
gold = load_golden("pcnet_default.npz")
net, _ = t.fresh_net(gold)
with torch.no_grad():
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.zero_(); m.running_var.zero_()
x, seq, labels = t.big_case(32, 76, 0)
with torch.no_grad():
    out = net(x.to(t.DEV), seq.to(t.DEV))
d = {k: v.detach().cpu() for k, v in net.state_dict().items() if "running" in k}
for i, o in enumerate(out):
    d[f"out{i}"] = o.detach().cpu()
if len(sys.argv) > 2:
    ref = torch.load(sys.argv[1])
    for k in d:
        same = torch.equal(d[k], ref[k])
        print(f"  {'same bits' if same else 'DIFFERENT'}  max|diff| {float((d[k].double() - ref[k].double()).abs().max()):.3e}  {k}")
else:
    torch.save(d, sys.argv[1])
    print("saved", sys.argv[1])
