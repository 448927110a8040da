#!/usr/bin/env python3
"""Engine 5 (streaming MFMA cascade) against engine 2 (exact f32), per octave and per frame range; then timing of the CQT stage at 256 clips."""
import os, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np, torch
import ake_amd
from ake_amd.cqt import CQTPlan
DEV = "cuda:0"
g = torch.Generator().manual_seed(11)
cases = (("full", 3, 330750, 22050, 4410, 288, 36), ("b17", 17, 30000, 22050, 4410, 288, 36), ("tiny", 2, 5000, 22050, 4410, 288, 36),
         ("hop2205", 3, 40000, 22050, 2205, 288, 36), ("seven_octaves", 2, 60000, 11025, 2205, 252, 36), ("six_octaves", 2, 30000, 5512, 2300, 216, 36),
         ("b40_odd_stride", 40, 44113, 22050, 4411, 288, 36))
for name, B, n, sr, hop, bins, bpo in cases:
    y = (torch.rand((B, n), generator=g) * 2 - 1).to(DEV)
    ref = CQTPlan(sr, hop, bins, bpo, engine=2).logmag(y).cpu().numpy()
    got = CQTPlan(sr, hop, bins, bpo, engine=5).logmag(y).cpu().numpy()
    peak = np.abs(ref).max()
    errs = []
    for o in range(bins // bpo):
        k0 = bins - bpo * (o + 1)
        d = np.abs(got[:, k0:k0 + bpo] - ref[:, k0:k0 + bpo])
        errs.append(float(d.max() / peak))
    T = ref.shape[2]
    d = np.abs(got - ref).max(axis=(0, 1)) / peak
    print(f"{name:16s} finite {np.isfinite(got).all()}  per-octave err " + " ".join(f"{e:.1e}" for e in errs) + f"   worst frames {np.argsort(-d)[:4].tolist()} of {T}", flush=True)
if len(sys.argv) > 1:
    from ake_amd import synthetic
    audio = synthetic.make_batch_device(range(256), torch.device(DEV))[0]
    for eng in (3, 5):
        plan = CQTPlan(22050, 4410, 288, 36, engine=eng)
        for _ in range(5): plan.logmag(audio)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): plan.logmag(audio)
        torch.cuda.synchronize(); print(f"engine {eng}: {(time.perf_counter() - t0) / 20 * 1e3:.4f} ms per 256 clips (incl. transpose)")
        ake_amd._lib.prof_enable("", True)
        for _ in range(10): plan.logmag(audio)
        torch.cuda.synchronize()
        print({k: round(v[0] / 10, 4) for k, v in ake_amd._lib.prof_results().items()})
        ake_amd._lib.prof_enable("", False)
