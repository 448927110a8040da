#!/usr/bin/env python3
"""CPU prototype: what the network outputs lose when the convolutions' ACTIVATION operand is rounded to one 16-bit float instead
of the hi + lo pair the kernels use now (weights stay hi + lo, i.e. exact to 2^-17 / 2^-22).  The oracle's float64 forward with a
rounding hook on every convolution input; prints max|out - ref| / max|ref| per output.
    python3 tests/tools/split_precision_proto.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden_state_dict  # noqa: E402
from oracle import pcnet_oracle  # noqa: E402

gold = np.load(os.path.join(ROOT, "tests/golden/pcnet_default.npz"), allow_pickle=False)
sd = golden_state_dict(gold, torch.float64)
real_conv = torch.nn.functional.conv2d


def rnd(x, kind):
    if kind == "f16":
        return x.to(torch.float16).to(torch.float64)
    if kind == "bf16":
        return x.to(torch.float32).to(torch.bfloat16).to(torch.float64)
    if kind == "bf16x2":                                   # hi + lo
        hi = x.to(torch.float32).to(torch.bfloat16).to(torch.float64)
        return hi + (x - hi).to(torch.float32).to(torch.bfloat16).to(torch.float64)
    if kind == "f16x2":
        hi = x.to(torch.float16).to(torch.float64)
        return hi + (x - hi).to(torch.float16).to(torch.float64)
    return x


def run(x, seq, act_kind, w_kind, only=None):
    def conv(inp, w, b=None, *a, **k):
        sel = only is None or only(w)
        return real_conv(rnd(inp, act_kind) if sel else inp, rnd(w, w_kind) if sel else w, b, *a, **k)
    pcnet_oracle.F.conv2d = conv
    try:
        with torch.no_grad():
            return pcnet_oracle.pcnet_forward(sd, x, seq)
    finally:
        pcnet_oracle.F.conv2d = real_conv


g = torch.Generator().manual_seed(0)
cases = [("golden input", torch.from_numpy(gold["x"]).double(), torch.from_numpy(gold["seq_length"]))]
# the bench's clips: sine mixes -> sparse log-CQT maps
t = torch.arange(76, dtype=torch.float64)
xs = torch.zeros((8, 1, 288, 76), dtype=torch.float64)
for b in range(8):
    for _ in range(6):
        p = int(torch.randint(0, 288, (1,), generator=g))
        xs[b, 0, max(p - 1, 0):p + 2, :] += torch.rand(1, generator=g).double() * 3
cases.append(("sparse maps", xs + 0.01 * torch.rand(xs.shape, generator=g).double(), None))
is_p2p = lambda w: w.shape[2] == 7 and w.shape[3] == 7 and w.shape[0] == 8
groups = {
    "pitch convs (7x7, 8 ch)": is_p2p,
    "semitone convs (3x3 stride 3)": lambda w: w.shape[2] == 3 and w.shape[3] == 3,
    "layer-0 pitch-class stack (4 ch)": lambda w: w.shape[2] == 12 and w.shape[0] == 4,
    "layer-1 pitch-class stack (16 ch)": lambda w: w.shape[2] == 12 and w.shape[0] == 16,
    "head conv 0 (16 -> 32)": lambda w: w.shape[2] == 12 and w.shape[0] == 32,
    "head conv 1 (32 -> 1)": lambda w: w.shape[2] == 12 and w.shape[0] == 1,
    "genre head": lambda w: w.shape[2] in (1, 2) and w.shape[3] == 7,
    "all": None,
}
for name, x, seq in cases:
    ref = run(x, seq, None, None)
    print(name)
    import sys as _s
    combos = (("f16", "f16x2"), ("f16", "f16"), ("bf16x2", "bf16x2")) if len(_s.argv) < 2 else tuple(tuple(None if v == "exact" else v for v in c.split(":")) for c in _s.argv[1:])
    for ak, wk in combos:
        for gname, only in groups.items():
            out = run(x, seq, ak, wk, only)
            errs = [float((o - r).abs().max() / r.abs().max()) for o, r in zip(out, ref)]
            print(f"   act {str(ak):7s} w {str(wk):7s} {gname:36s} key {errs[0]:.2e}  tonic {errs[1]:.2e}  genre {errs[2]:.2e}")
