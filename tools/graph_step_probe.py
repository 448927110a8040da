#!/usr/bin/env python3
"""Does a training step (train-mode forward + loss + HIP backward) survive torch.cuda.graph capture, and what does replay save?
python3 tools/graph_step_probe.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import numpy as np, torch
import ake_amd
from ake_amd import synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "pcnet_default.npz"))
sd = {k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd/")}

def make():
    net = ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(genre=True, lr=3e-4, gamma=0.96, acc_grad=1)); net.load_state_dict(sd); net = net.to(dev).train()
    net.trainer = ake_amd.Trainer(accumulate_grad_batches=1)
    return net, net.configure_optimizers()[0][0]

audio, labels = synthetic.make_batch_device(range(B), dev, n_samples=22050 * 15)
mel = ake_amd.cqt_logmag(audio, 22050, 4410, n_bins=288, bins_per_octave=36)[:, None].contiguous()
batch = {"mel": mel, "seq_length": torch.full((B,), mel.shape[-1], device=dev), **{k: torch.as_tensor(v).to(dev) for k, v in labels.items()}}

# eager reference
net, optim = make()
losses_e = []
for i in range(6):
    optim.zero_grad(); l = net.training_step(batch, i)["loss"]; l.backward(); optim.step(); losses_e.append(float(l))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10):
    optim.zero_grad(); net.training_step(batch, i)["loss"].backward(); optim.step()
torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 10

# graphed
net, optim = make()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
losses_g = []
with torch.cuda.stream(side):
    for i in range(3):
        optim.zero_grad(); l = net.training_step(batch, i)["loss"]; l.backward(); optim.step(); losses_g.append(float(l))
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    net._flat_grad.zero_()
    static_loss = net.training_step(batch, 0)["loss"]
    static_loss.backward()
for i in range(3):
    g.replay(); optim.step(); losses_g.append(float(static_loss))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10):
    g.replay(); optim.step()
torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 10
print("eager losses ", ["%.6f" % x for x in losses_e])
print("graph losses ", ["%.6f" % x for x in losses_g])
print(f"B={B}: eager {te*1e3:.3f} ms/step, graphed {tg*1e3:.3f} ms/step")
