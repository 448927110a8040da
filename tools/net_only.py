#!/usr/bin/env python3
"""PitchClassNet inference alone on 256 resident log-CQT clips (for rocprofv3 counter passes):  python3 tools/net_only.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import numpy as np, torch
import ake_amd
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "pcnet_default.npz"))
sd = {k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd/")}
net = ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(genre=True)); net.load_state_dict(sd); net = net.cuda().eval()
mel = torch.rand(256, 1, 288, 76, device="cuda") * 2.5
seq = torch.full((256,), 76, device="cuda")
for _ in range(steps):
    out = net(mel, seq)
torch.cuda.synchronize()
print(float(out[0].mean()))
