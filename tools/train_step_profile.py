"""Where does a training step's wall time go?  (host-side phases with device syncs; GPU box only)"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import numpy as np, torch
import ake_amd, bench
from ake_amd import synthetic
dev = torch.device("cuda", 0)
B = int(os.environ.get("B", 256))
net = ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(genre=True, lr=3e-4, gamma=0.96, acc_grad=1))
net.load_state_dict(bench.load_fixture_weights()); net = net.to(dev).train()
audio, labels = synthetic.make_batch_device(range(B), dev)
mel = ake_amd.cqt_logmag(audio, 22050, 4410, n_bins=288, bins_per_octave=36)[:, None].contiguous()
batch = {"mel": mel, "seq_length": torch.full((B,), 76, device=dev), **{k: torch.as_tensor(v).to(dev) for k, v in labels.items()}}
optim = net.configure_optimizers()[0][0]
net.trainer = ake_amd.Trainer()
def sync(): torch.cuda.synchronize(); return time.perf_counter()
def step(i, timing=None):
    t0 = sync(); optim.zero_grad()
    out = net.training_step(batch, i); t1 = sync()
    out["loss"].backward(); t2 = sync()
    optim.step(); t3 = sync()
    if timing is not None: timing.append((t1 - t0, t2 - t1, t3 - t2))
for i in range(3): step(i)
tm = []
for i in range(10): step(i, tm)
print("ms: fwd+loss %.2f  bwd %.2f  adam %.2f" % tuple(1e3 * np.mean(tm, 0)))
t0 = sync(); key, tonic, genre = net(batch["mel"], batch["seq_length"]); t1 = sync()
print("forward only (autograd node) ms: %.2f" % (1e3 * (t1 - t0)))
pr = cProfile.Profile(); pr.enable()
for i in range(5): step(i)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
