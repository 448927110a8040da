#!/usr/bin/env python3
"""CQT stage alone on 256 resident clips (for rocprofv3 counter passes):  python3 tools/cqt_only.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ake_amd
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
audio = torch.rand(256, 330750, device="cuda") - 0.5
for _ in range(steps):
    out = ake_amd.cqt_logmag(audio, 22050, 4410, n_bins=288, bins_per_octave=36)
torch.cuda.synchronize()
print(out.shape, float(out.mean()))
