#!/usr/bin/env python3
"""GPU debug harness for CQT engine 4 (fused cascade + bank): per-octave error against engine 2 (exact-f32 multirate) for a
set of shapes, then a timing of the bench batch.   python3 tests/tools/cqt_engine4_check.py [--time]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ake_amd  # noqa: E402
from ake_amd.cqt import CQTPlan  # noqa: E402

DEV = "cuda:0"
CASES = (("full_b3", 3, 330750, 22050, 4410, 288, 36), ("short_b17", 17, 30000, 22050, 4410, 288, 36), ("tiny", 2, 5000, 22050, 4410, 288, 36),
         ("hop2205", 3, 40000, 22050, 2205, 288, 36), ("oct7", 2, 60000, 11025, 2205, 252, 36), ("oct4", 2, 20000, 1378, 2205, 144, 36),
         ("oct6", 2, 30000, 5512, 2300, 216, 36), ("oct1", 2, 3000, 172, 300, 36, 36), ("b40_odd", 40, 44100 + 13, 22050, 4411, 288, 36))


def main():
    g = torch.Generator().manual_seed(11)
    bad = 0
    for name, B, n, sr, hop, bins, bpo in CASES:
        y = (torch.rand((B, n), generator=g) * 2 - 1).to(DEV)
        ref = CQTPlan(sr, hop, bins, bpo, engine=2).logmag(y).cpu().numpy()
        got = CQTPlan(sr, hop, bins, bpo, engine=4).logmag(y).cpu().numpy()
        n_oct = bins // bpo
        peak = np.abs(ref).max()
        errs = []
        for o in range(n_oct):
            k0 = bins - bpo * (o + 1)
            e = np.abs(got[:, k0:k0 + bpo] - ref[:, k0:k0 + bpo])
            errs.append(float(e.max() / peak))
        worst = max(errs)
        print(f"{name:10s} B={B:3d} n={n:6d} T={ref.shape[2]:3d}  finite={np.isfinite(got).all()}  rel err per octave (top first): "
              + " ".join(f"{e:.1e}" for e in errs), flush=True)
        if not worst < 5e-5:
            bad += 1
            o = int(np.argmax(errs))
            k0 = bins - bpo * (o + 1)
            e = np.abs(got[:, k0:k0 + bpo] - ref[:, k0:k0 + bpo]).max(axis=1)          # (B, T)
            print("   worst octave", o, "max err per frame (clip 0):", np.array2string(e[0] / peak, precision=1, max_line_width=250))
            print("   per clip:", np.array2string(e.max(axis=1) / peak, precision=1, max_line_width=250))
    # ragged
    lens = [70001, 70000, 4410 * 3 + 7, 50000, 4409, 30000]
    rows = torch.full((len(lens), max(lens) + 5), float("nan"))
    for i, n in enumerate(lens):
        rows[i, :n] = torch.rand(n, generator=g) * 2 - 1
    audio = rows.to(DEV)[:, :max(lens)]
    p3, p4 = CQTPlan(22050, 4410, 288, 36, engine=3), CQTPlan(22050, 4410, 288, 36, engine=4)
    a = p3.logmag(audio, lengths=torch.tensor(lens)).cpu().numpy()
    b = p4.logmag(audio, lengths=torch.tensor(lens)).cpu().numpy()
    print("ragged: finite", np.isfinite(b).all(), "rel err vs engine 3", float(np.abs(a - b).max() / np.abs(a).max()))
    if "--time" in sys.argv:
        from ake_amd import synthetic
        audio, _ = synthetic.make_batch_device(range(256), torch.device(DEV))
        runs = [(3, "0"), (4, "0")] + ([(4, d) for d in ("1", "3", "7")] if "--dbg" in sys.argv else [])
        for eng, dbg in runs:
            os.environ["AKE_CQT_FZ_DBG"] = dbg
            p = CQTPlan(22050, 4410, 288, 36, engine=eng)
            for _ in range(3):
                p.logmag(audio)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                p.logmag(audio)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            ake_amd._lib.prof_enable("", True)
            for _ in range(5):
                p.logmag(audio)
            res = ake_amd._lib.prof_results()
            ake_amd._lib.prof_enable("", False)
            print(f"engine {eng} dbg {dbg}: {dt * 1e3:.3f} ms per 256 clips;", {k: round(v[0] / 5, 4) for k, v in res.items()}, flush=True)
    print("BAD CASES:", bad)


if __name__ == "__main__":
    main()
