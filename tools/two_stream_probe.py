#!/usr/bin/env python3
"""Does issuing independent steps on two streams (own net handle, workspace and outputs each) overlap the VALU/HBM-bound CQT of
one step with the MFMA-bound network of another?  python3 tools/two_stream_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import numpy as np, torch
import ake_amd
from ake_amd import synthetic

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "pcnet_default.npz"))
sd = {k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd/")}
dev = torch.device("cuda", 0)
ests = []
for _ in range(2):
    net = ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(genre=True)); net.load_state_dict(sd); net = net.to(dev).eval()
    ests.append(ake_amd.KeyEstimator(net, 22050, 5))
audio, _ = synthetic.make_batch_device(range(256), dev)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for e, s in zip(ests, streams):
    with torch.cuda.stream(s):
        for _ in range(3): e(audio)
torch.cuda.synchronize()

def run(n_streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        j = i % n_streams
        with torch.cuda.stream(streams[j]):
            ests[j](audio)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps

for rep in range(2):
    a, b = run(1), run(2)
    print(f"1 stream {a * 1e3:.3f} ms/step ({256 / a:.0f} clips/s)   2 streams {b * 1e3:.3f} ms/step ({256 / b:.0f} clips/s)")
