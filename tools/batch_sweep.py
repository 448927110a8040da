#!/usr/bin/env python3
"""Latency / throughput of the whole hot path (audio -> key, tonic, genre) over the batch size: python3 tools/batch_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from argparse import Namespace
import ake_amd
from ake_amd import synthetic
gold = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "pcnet_default.npz"))
sd = {k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd/")}
net = ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(genre=True)); net.load_state_dict(sd); net = net.cuda().eval()
est = ake_amd.KeyEstimator(net)
for B in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024):
    audio = torch.rand(B, synthetic.N_SAMPLES, device="cuda") - 0.5
    for _ in range(3): est(audio)
    torch.cuda.synchronize()
    n = 50 if B <= 64 else 20
    t0 = time.perf_counter()
    for _ in range(n): est(audio)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"B={B:5d}  {dt*1e3:8.3f} ms per call  {B/dt:10.0f} clips/s", flush=True)
