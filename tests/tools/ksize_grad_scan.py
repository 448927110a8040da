#!/usr/bin/env python3
"""Scan (kernel_size, frames, seed) cases of tests/test_gpu_backward.py::test_kernel_size_gradients for kink-free picks:
    python3 tests/tools/ksize_grad_scan.py"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
from argparse import Namespace
import torch
import ake_amd
from test_gpu_backward import make_case, reference_grads, loss_fn, grad_errors, ILL_CONDITIONED, DEV
for ksz in (3, 5):
    for frames in (40, 52):
        for seed in range(6):
            opt = Namespace(conv_layers=3, n_filters=4, head_layers=2, time_pool_size=2, genre=True, max_pool=False, frames=5, kernel_size=ksz)
            torch.manual_seed(50 + seed)
            net = ake_amd.PitchClassNet(288, 12, 2, ksz, opt)
            sd32 = {k: v.clone() for k, v in net.state_dict().items()}
            x, seq, labels = make_case(2, frames, seed)
            loss_ref, ref = reference_grads(sd32, x, seq, labels, kernel_size=ksz)
            net = net.to(DEV).train()
            out = net(x.to(DEV), seq.to(DEV))
            loss = loss_fn(out[0], out[1], out[2], *(t.to(DEV) for t in labels))
            loss.backward()
            rows = grad_errors(net, ref)
            tight = [r for r in rows if r[1] not in ILL_CONDITIONED]
            print(f"k={ksz} T={frames} seed={seed}: loss err {abs(float(loss.detach()) - loss_ref):.1e}  worst {tight[0][0]:.2e} ({tight[0][1]})  median {rows[len(rows)//2][0]:.2e}", flush=True)
