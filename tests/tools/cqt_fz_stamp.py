#!/usr/bin/env python3
"""In-kernel cycle stamps of CQT engine 4's step loop (diagnostic build, AKE_CQT_FZ_STAMP=1): shares, not run times."""
import os
import sys
os.environ["AKE_CQT_FZ_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from ake_amd import synthetic  # noqa: E402
from ake_amd.cqt import CQTPlan  # noqa: E402

audio, _ = synthetic.make_batch_device(range(256), torch.device("cuda:0"))
p = CQTPlan(22050, 4410, 288, 36, engine=4)
for dbg in sys.argv[1:] or ["0"]:
    os.environ["AKE_CQT_FZ_DBG"] = dbg
    print("dbg", dbg, file=sys.stderr, flush=True)
    p.logmag(audio)
    torch.cuda.synchronize()
    p.logmag(audio)
    torch.cuda.synchronize()
