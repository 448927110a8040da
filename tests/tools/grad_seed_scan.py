"""Scan (shape, seed) cases of the gradient parity test: worst per-tensor error of the HIP backward vs float64 autograd
through the oracle.  Cases above ~1e-5 are LeakyReLU kink flips (one pre-activation whose sign differs between the f32
forward and the f64 forward), see tests/test_gpu_backward.py."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import numpy as np, torch
import ake_amd
import test_gpu_backward as tb

gold = np.load("tests/golden/pcnet_default.npz")
sd32 = {k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd/")}
opt = Namespace(**json.loads(str(gold["opt"])))
net = ake_amd.PitchClassNet(288, 12, 2, 7, opt); net.load_state_dict(sd32); net = net.cuda().train()
shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("SHAPES", "4x40,4x52,2x76,3x64").split(",")]
for batch, frames in shapes:
    for seed in range(int(os.environ.get("SEEDS", 6))):
        net.zero_grad(set_to_none=True)
        x, seq, labels = tb.make_case(batch, frames, seed)
        _, ref = tb.reference_grads(sd32, x, seq, labels)
        out = net(x.cuda(), seq.cuda())
        tb.loss_fn(out[0], out[1], out[2], *(t.cuda() for t in labels)).backward()
        rows = tb.grad_errors(net, ref)
        print(f"B={batch} T={frames} seed={seed}: worst {rows[0][0]:.2e} ({rows[0][1]}), median {rows[len(rows)//2][0]:.2e}, tensors>1e-5: {sum(r[0] > 1e-5 for r in rows)}", flush=True)
