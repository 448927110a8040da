#!/usr/bin/env python3
"""In-kernel cycle stamps of engine 3's decimator cascade (diagnostic build, AKE_CQT_CASC_STAMP=1): cycles per phase of a tick."""
import os
import sys
os.environ["AKE_CQT_CASC_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from ake_amd import synthetic  # noqa: E402
from ake_amd.cqt import CQTPlan  # noqa: E402

audio, _ = synthetic.make_batch_device(range(256), torch.device("cuda:0"))
p = CQTPlan(22050, 4410, 288, 36, engine=3)
for _ in range(2):
    p.logmag(audio)
    torch.cuda.synchronize()
