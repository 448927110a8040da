#!/usr/bin/env python3
"""One training forward + backward of the default net at the reference's batch size (8 clips, 76 frames): run with AKE_DEBUG=1 to print the
tile geometry every generic convolution picks (choose_tile / want_tiles), see DESIGN.md section 5 "BASELINE configs[2]"."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch, ake_amd
opt = Namespace(conv_layers=3, n_filters=4, head_layers=2, time_pool_size=2, genre=True, max_pool=False, frames=5, octaves=8, lr=3e-4)
torch.manual_seed(0)
net = ake_amd.PitchClassNet(288, 12, 2, 7, opt).cuda().train()
x = torch.rand(8, 1, 288, 76, device="cuda") * 2.5
seq = torch.full((8,), 76, device="cuda")
out = net(x, seq)
(out[0].sum() + out[1].sum() + out[2].sum()).backward()
torch.cuda.synchronize()
