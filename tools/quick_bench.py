#!/usr/bin/env python3
"""Per-kernel ms/step of the hot path at B=256 (hipEvent timer), for A/B experiments:  python3 tools/quick_bench.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from argparse import Namespace
import ake_amd
from ake_amd import synthetic
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
gold = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "pcnet_default.npz"))
sd = {k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd/")}
net = ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(genre=True)); net.load_state_dict(sd); net = net.cuda().eval()
est = ake_amd.KeyEstimator(net)
audio = torch.rand(256, synthetic.N_SAMPLES, device="cuda") - 0.5
for _ in range(2): est(audio)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(steps): est(audio)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
ake_amd._lib.prof_enable("", True)
for _ in range(steps): est(audio)
res = ake_amd._lib.prof_results(); ake_amd._lib.prof_enable("", False)
print(f"ms/step {dt*1e3:.3f}  clips/s {256/dt:.0f}  " + "  ".join(f"{k.split('/')[-1]}={v[0]/steps:.3f}" for k, v in sorted(res.items(), key=lambda kv: -kv[1][0])))
