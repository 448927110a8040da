#!/usr/bin/env python3
"""Timing of cqt_stream_kernel with phases switched off (diagnostic build: AKE_USE_DIAG_LIB=1 AKE_SM_ABLATE=bits, AKE_CQT_SEGS=n)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ake_amd
from ake_amd.cqt import CQTPlan
from ake_amd import synthetic
audio = synthetic.make_batch_device(range(256), torch.device("cuda:0"))[0]
plan = CQTPlan(22050, 4410, 288, 36, engine=5)
for _ in range(5): plan.logmag(audio)
ake_amd._lib.prof_enable("", True)
for _ in range(20): plan.logmag(audio)
torch.cuda.synchronize()
print({k: v for k, v in os.environ.items() if k.startswith("AKE_")}, {k: round(v[0] / 20, 4) for k, v in ake_amd._lib.prof_results().items()})
