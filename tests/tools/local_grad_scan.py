"""Diagnostic: per-tensor gradient errors of the --local training path for a few (shape, seed) cases (kink flips vs systematic)."""
import json, sys
from argparse import Namespace
import numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ake_amd
from conftest import golden_state_dict
from oracle import pcnet_oracle
from test_gpu_backward import _local_loss, grad_errors

gold = np.load(os.path.join(ROOT, "tests/golden/pcnet_default.npz"), allow_pickle=False)
DEV = "cuda:0"
CASES = ([tuple(int(v) for v in c.split("x")) for c in os.environ["CASES"].split(",")] if os.environ.get("CASES") else
         [(3, 120, sd_) for sd_ in range(8)] + [(2, 150, sd_) for sd_ in range(6)] + [(1, 300, 0), (2, 90, 1)])
for batch, frames, seed in CASES:
    opt = Namespace(**json.loads(str(gold["opt"])))
    opt.local = True
    net = ake_amd.PitchClassNet(288, 12, 2, 7, opt)
    sd32 = golden_state_dict(gold)
    net.load_state_dict(sd32, strict=True)
    net = net.to(DEV).train()
    W = net.local_window
    g = torch.Generator().manual_seed(seed)
    x = torch.rand((batch, 1, 288, frames), generator=g) * 2.5
    Tm = frames - 12; Tq = Tm - W + 1
    ns = [Tq, Tq - 9, Tq - 20][:batch]
    kl = (torch.rand((batch, Tq, 12), generator=g) > 0.5).float()
    ti = torch.randint(0, 12, (batch, Tq), generator=g)
    gi = torch.randint(0, 11, (batch, Tm), generator=g)
    sd = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.double() if v.is_floating_point() else v)
          for k, v in sd32.items()}
    out = pcnet_oracle.pcnet_forward(sd, x.double(), None, training=True, local_window=W)
    _local_loss(out[0], out[1], out[2], kl, ti, gi, ns).backward()
    ref = {k: v.grad for k, v in sd.items() if torch.is_tensor(v) and v.requires_grad}
    o = net(x.to(DEV), None)
    _local_loss(o[0], o[1], o[2], kl.to(DEV), ti.to(DEV), gi.to(DEV), ns).backward()
    rows = grad_errors(net, ref)
    by = {n: e for e, n, _ in rows}
    print(f"case B={batch} T={frames} seed={seed}: worst {rows[0][0]:.2e} median {rows[len(rows)//2][0]:.2e}")
    if "-v" in sys.argv:
        for n, _ in net.named_parameters():
            if n in by:
                print(f"    {by[n]:9.2e} {n}")
