#!/usr/bin/env python3
"""--denseblock training: loss and gradients against float64 autograd through the oracle, a few (widths, frames, seed) cases:
    python3 tests/tools/dense_grad_scan.py [verbose]"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
from argparse import Namespace
import torch
import ake_amd
from test_gpu_backward import make_case, reference_grads, loss_fn, grad_errors, DEV
verbose = len(sys.argv) > 1
L = int(os.environ.get("DENSE_LAYERS", "2"))
for nf, cl in (((2, 2), (4, 3)) if L < 3 else ((1, 1), (2, 2))):
    for frames in ((40, 52) if L < 3 else (96,)):
        for seed in range(3 if L < 3 else 6):
            opt = Namespace(conv_layers=cl, n_filters=nf, head_layers=2, time_pool_size=2, genre=True, max_pool=False, frames=5, denseblock=True)
            torch.manual_seed(60 + seed)
            net = ake_amd.PitchClassNet(288, 12, L, 7, opt)
            sd32 = {k: v.clone() for k, v in net.state_dict().items()}
            x, seq, labels = make_case(2, frames, seed)
            loss_ref, ref = reference_grads(sd32, x, seq, labels)
            net = net.to(DEV).train()
            out = net(x.to(DEV), seq.to(DEV))
            loss = loss_fn(out[0], out[1], out[2], *(t.to(DEV) for t in labels))
            loss.backward()
            rows = grad_errors(net, ref)
            print(f"L={L} nf={nf} layers={cl} T={frames} seed={seed}: loss err {abs(float(loss.detach()) - loss_ref):.1e}  worst {rows[0][0]:.2e} ({rows[0][1]})  median {rows[len(rows)//2][0]:.2e}", flush=True)
            if verbose:
                for e, nme, m in rows[:40]:
                    print(f"     {e:9.2e}  max|ref|={m:9.2e}  {nme}")
                sys.exit(0)
