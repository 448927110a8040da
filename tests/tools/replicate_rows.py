#!/usr/bin/env python3
"""Per-tensor difference between the gradients of 8 shuffled copies of a 32-clip batch and those of the 32-clip batch itself (both on
the device; tests/test_gpu_train_scale.py).  AKE_USE_DIAG_LIB=1 + AKE_WGRAD_F32=1 / AKE_P2P_TRAIN_F32=1 / AKE_PC_TRAIN_F32=1 bisect."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import torch
from conftest import load_golden
import test_gpu_train_scale as t

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 8
gold = load_golden("pcnet_default.npz")
net, _ = t.fresh_net(gold)
x, seq, labels = t.big_case(32, 76, 0)
l32, g32 = t.device_grads(net, x, seq, labels)
x8, seq8, labels8 = t.replicate(x, seq, labels, copies, 5)
l256, g256 = t.device_grads(net, x8, seq8, labels8)
print("env:", {k: v for k, v in os.environ.items() if k.startswith("AKE_")}, "copies", copies, "loss", l32, l256)
rows = {r[1]: r for r in t.error_rows(g256, g32)}
for n, _ in net.named_parameters():
    if n in rows:
        e, _, m, u = rows[n]
        print(f"  {e:9.2e}  unfloored {u:9.2e}  max|ref| {m:9.2e}  {n}")
