#!/usr/bin/env python3
"""Time one architecture variant's inference forward (random weights, 256 clips x 76 frames): python3 tools/variant_bench.py <flag|default> [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch
import ake_amd

flag = sys.argv[1] if len(sys.argv) > 1 else "default"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
opt = Namespace(genre=True, **({} if flag == "default" else {flag: True}))
net = ake_amd.PitchClassNet(288, 12, 2, 7, opt).cuda().eval()
mel = torch.rand(B, 1, 288, 76, device="cuda") * 2.5
seq = torch.full((B,), 76, device="cuda")
for _ in range(2):
    net(mel, seq)
torch.cuda.synchronize()
ake_amd._lib.prof_enable("", True)
t0 = time.perf_counter()
steps = 5
for _ in range(steps):
    net(mel, seq)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
res = ake_amd._lib.prof_results()
ake_amd._lib.prof_enable("", False)
print(f"{flag}: {dt * 1e3:.2f} ms per {B} clips = {B / dt:.0f} clips/s (with the per-launch event timer on)")
print("  " + "  ".join(f"{k}={v[0] / steps:.3f}ms/{v[1] // steps}" for k, v in sorted(res.items(), key=lambda kv: -kv[1][0])[:8]))
