#!/usr/bin/env python3
"""Per-tensor gradient errors of the default net at a large batch against float64 autograd through the oracle (the statement of
tests/test_gpu_train_scale.py, every row printed).  Environment switches of the library (AKE_P2P_TRAIN_F32=1, AKE_PC_TRAIN_F32=1,
AKE_WGRAD_F32=1) select the f32 kernels for bisecting.      python3 tests/tools/big_grad_rows.py [batch=32] [seed=1] [frames=76]"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import torch
from conftest import load_golden
import test_gpu_train_scale as t

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 76
gold = load_golden("pcnet_default.npz")
net, sd32 = t.fresh_net(gold)
x, seq, labels = t.big_case(B, frames, seed)
loss_ref, ref = t.oracle_grads(sd32, x, seq, labels)
loss, got = t.device_grads(net, x, seq, labels)
print("env:", {k: v for k, v in os.environ.items() if k.startswith("AKE_")}, " batch", B, "seed", seed, " loss", loss, loss_ref)
order = [n for n, _ in net.named_parameters()]
rows = {r[1]: r for r in t.error_rows(got, ref)}
for n in order:
    if n in rows:
        e, _, m, u = rows[n]
        print(f"  {e:9.2e}  max|ref| {m:9.2e}  {n}")
