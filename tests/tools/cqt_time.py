#!/usr/bin/env python3
"""Per-kernel time of the CQT stage at 256 clips.   python3 tests/tools/cqt_time.py [engine=3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ake_amd
from ake_amd.cqt import CQTPlan
from ake_amd import synthetic
audio = synthetic.make_batch_device(range(256), torch.device("cuda:0"))[0]
plan = CQTPlan(22050, 4410, 288, 36, engine=int(sys.argv[1]) if len(sys.argv) > 1 else 3)
for _ in range(10): plan.logmag(audio)
ake_amd._lib.prof_enable("", True)
for _ in range(30): plan.logmag(audio)
torch.cuda.synchronize()
r = {k: round(v[0] / 30, 4) for k, v in ake_amd._lib.prof_results().items()}
print({k: v for k, v in os.environ.items() if k.startswith("AKE_")}, r, "cqt sum", round(sum(v for k, v in r.items() if "transpose" not in k), 4))
