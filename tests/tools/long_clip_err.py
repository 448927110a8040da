#!/usr/bin/env python3
"""Long clips (time-tiled kernels) against the float64 oracle, default and --local nets: python3 tests/tools/long_clip_err.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from argparse import Namespace
import numpy as np, torch
import ake_amd
from oracle import pcnet_oracle
gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden", "pcnet_default.npz"))
sd = {k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd/")}
sd64 = pcnet_oracle.to_dtype(sd, torch.float64)
def rel(a, b): return float((a.double().cpu() - b).abs().max() / b.abs().max())
for local in (False, True):
    net = ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(genre=True, local=local)); net.load_state_dict(sd); net = net.cuda().eval()
    for T in (200, 300, 500, 700, 1000, 1500):
        g = torch.Generator().manual_seed(T)
        x = torch.rand((1, 1, 288, T), generator=g) * 2.5
        ref = pcnet_oracle.pcnet_forward(sd64, x.double(), None, local_window=38 if local else None)
        got = net(x.cuda(), None)
        errs = [rel(a, b) for a, b in zip(got, ref)]
        extra = ""
        if local:
            d = (got[1].double().cpu() - ref[1]).abs().reshape(-1)
            bad = torch.nonzero(d > 1e-4 * ref[1].abs().max()).reshape(-1)
            extra = f" first bad flat index {int(bad[0]) if len(bad) else -1} of {d.numel()}"
        print(f"local={local} T={T}: {['%.1e' % e for e in errs]}{extra}", flush=True)
