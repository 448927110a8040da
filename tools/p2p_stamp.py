#!/usr/bin/env python3
"""In-kernel cycle stamps of the persistent pitch convolution's tile loop (diagnostic build, AKE_P2P_STAMP=1): shares, not run times.
    python3 tools/p2p_stamp.py"""
import os
import sys
from argparse import Namespace
os.environ["AKE_P2P_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import ake_amd  # noqa: E402

gold = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pcnet_default.npz"))
sd = {k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd/")}
net = ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(genre=True))
net.load_state_dict(sd, strict=True)
net = net.to("cuda:0").eval()
mel = torch.rand((256, 1, 288, 76), device="cuda:0") * 2.5
seq = torch.full((256,), 76, device="cuda:0")
for _ in range(3):
    net(mel, seq)
    torch.cuda.synchronize()
