#!/usr/bin/env python3
"""Which activation first departs from the oracle on a long clip: python3 tests/tools/long_clip_bisect.py T [local]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from argparse import Namespace
import numpy as np, torch
import ake_amd
from oracle import pcnet_oracle
T = int(sys.argv[1]); local = len(sys.argv) > 2
gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden", "pcnet_default.npz"))
sd = {k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd/")}
sd64 = pcnet_oracle.to_dtype(sd, torch.float64)
net = ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(genre=True, local=local)); net.load_state_dict(sd); net = net.cuda().eval()
net.keep_taps(True)
x = torch.rand((1, 1, 288, T), generator=torch.Generator().manual_seed(T)) * 2.5
taps = {}
pcnet_oracle.pcnet_forward(sd64, x.double(), None, local_window=38 if local else None, taps=taps)
net(x.cuda(), None)
for name in ["model.0.pool", "model.0.pc2pc.layer.2", "model.0.pc2pc.layer.5", "model.1.up_sixth_a", "model.1.p2p.layer.5", "model.1.p2p.layer.8", "model.1.cat",
             "model.1.pc2pc.layer.5", "model.1.pc2pc.layer.8", "model.1.time_pool_pc", "key_map", "tonic_map", "genre_map"]:
    try:
        got = net.tap(name).double().cpu()
    except Exception as e:
        print(name, "tap error", str(e)[:80]); continue
    if name == "model.1.cat":
        ref = torch.cat([taps["model.0.pc2pc.layer.8"], taps["model.1.pool"]], 1)
    else:
        ref = taps[name]
    d = (got - ref).abs()
    bad_t = torch.nonzero(d.amax(dim=(0, 1, 2)) > 1e-4 * ref.abs().max()).reshape(-1)
    print(f"{name:28s} shape {tuple(got.shape)} rel {float(d.max() / ref.abs().max()):.1e}  bad frames: {bad_t[:6].tolist()}..{bad_t[-3:].tolist() if len(bad_t) else ''} ({len(bad_t)})")
