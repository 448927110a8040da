#!/usr/bin/env python3
"""Time the pitch-convolution launches alone (hipEvent timer): python3 tools/p2p_only.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import numpy as np, torch
import ake_amd
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "pcnet_default.npz"))
sd = {k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd/")}
net = ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(genre=True)); net.load_state_dict(sd); net = net.cuda().eval()
mel = torch.rand(256, 1, 288, 76, device="cuda") * 2.5
seq = torch.full((256,), 76, device="cuda")
for _ in range(3): net(mel, seq)
torch.cuda.synchronize()
ake_amd._lib.prof_enable("conv_p2p", True)
for _ in range(steps): net(mel, seq)
torch.cuda.synchronize()
res = ake_amd._lib.prof_results(); ake_amd._lib.prof_enable("", False)
print("  ".join(f"{k}={v[0]/steps:.4f}ms/{v[1]//steps}" for k, v in sorted(res.items())))
