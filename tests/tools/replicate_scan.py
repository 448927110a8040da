#!/usr/bin/env python3
"""Gradient difference (k shuffled copies of a 32-clip batch) - (the batch), selected tensors, k = 2..8."""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import torch
from conftest import load_golden
import test_gpu_train_scale as t
gold = load_golden("pcnet_default.npz")
net, _ = t.fresh_net(gold)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
x, seq, labels = t.big_case(32, 76, seed)
l32, g32 = t.device_grads(net, x, seq, labels)
names = ["model.1.pc2pc.layer.7.bias", "model.1.pc2pc.layer.1.bias", "model.1.pool_semi_b.bias", "model.1.p2p.layer.7.bias", "model.1.p2p.layer.6.weight", "model.1.p2p.layer.4.bias",
         "model.1.p2p.layer.3.weight", "model.1.p2p.layer.1.bias", "model.1.p2p.layer.0.weight", "model.1.up_sixth_b.bias", "model.0.pc2pc.layer.7.bias", "model.0.pool_semi.weight"]
print("copies  loss-equal  " + "  ".join(n.replace("model.", "m").replace("layer.", "l")[:14].ljust(14) for n in names))
for k in (1, 2, 3, 4, 5, 6, 8):
    xk, sk, lk = t.replicate(x, seq, labels, k, 5)
    lk_, gk = t.device_grads(net, xk, sk, lk)
    rows = {r[1]: r[0] for r in t.error_rows(gk, g32)}
    print(f"{k:5d}   {str(lk_ == l32):9s}  " + "  ".join(f"{rows[n]:14.2e}" for n in names), flush=True)
