#!/usr/bin/env python3
"""--denseblock training at 32 and 64 clips x 76 frames, default widths: finite gradients, step time (generic kernels).
    python3 tests/tools/dense_batch_probe.py"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from argparse import Namespace
import torch, ake_amd
opt = Namespace(conv_layers=3, n_filters=4, head_layers=2, time_pool_size=2, genre=True, max_pool=False, frames=5, denseblock=True)
torch.manual_seed(3)
net = ake_amd.PitchClassNet(288, 12, 2, 7, opt).cuda().train()
g = torch.Generator().manual_seed(4)
for B in (32, 64):
    x = (torch.rand((B, 1, 288, 76), generator=g) * 2.5).cuda()
    seq = torch.randint(60, 77, (B,), generator=g).cuda()
    net.zero_grad(set_to_none=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = net(x, seq)
    loss = out[0].mean() + out[1].mean() + out[2].mean()
    loss.backward()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    gs = [p.grad for p in net.parameters()]
    print(B, "loss", float(loss.detach()), "finite", all(torch.isfinite(t).all().item() for t in gs), "max grad", max(float(t.abs().max()) for t in gs), f"{dt*1e3:.1f} ms")
