#!/usr/bin/env python3
"""Inference forward and training step of every architecture variant the library runs, random weights, synthetic
log-CQT clips of 76 frames:  python3 tools/variant_sweep.py [batch=256]
The default net runs the specialised kernels; every variant runs the generic ones (conv_mfma_kernel, semi_fold_kernel, ...)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch
import ake_amd
from ake_amd import synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
_, labels = synthetic.make_batch_device(range(B), dev)
mel = torch.rand(B, 1, 288, 76, device=dev) * 2.5
seq = torch.full((B,), 76, device=dev)
VARIANTS = [("default", {}, 2), ("--local", {"local": True}, 2), ("--resblock", {"resblock": True}, 2), ("--denseblock", {"denseblock": True}, 2),
            ("--pc2p_mem", {"pc2p_mem": True}, 2), ("--p2pc_conv", {"p2pc_conv": True}, 2), ("--stay_sixth", {"stay_sixth": True}, 2),
            ("--kernel_size 3", {"kernel_size": 3}, 2), ("--kernel_size 5", {"kernel_size": 5}, 2), ("--num_layers 3", {}, 3)]


def timed(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


print(f"| variant | inference, ms per {B} clips | clips/s | training step (fwd + bwd + Adam), ms per {B} clips |")
print("|---|---|---|---|")
for name, kw, layers in VARIANTS:
    opt = Namespace(genre=not kw.get("local", False), lr=3e-4, gamma=0.96, acc_grad=1, **kw)
    ks = kw.get("kernel_size", 7)
    try:
        torch.manual_seed(0)
        net = ake_amd.PitchClassNet(288, 12, layers, ks, opt).to(dev).eval()
        with torch.no_grad():
            t_inf = timed(lambda: net(mel, seq), 5)
    except Exception as e:  # noqa: BLE001
        print(f"| {name} | refused: {str(e)[:80]} | | |", flush=True)
        continue
    t_tr = None
    try:
        net.train()
        optim = net.configure_optimizers()[0][0]
        net.trainer = ake_amd.Trainer()
        batch = {"mel": mel, "seq_length": seq, **{k: torch.as_tensor(v).to(dev) for k, v in labels.items()}}
        if kw.get("local"):
            raise NotImplementedError("per-frame labels: see tests/test_gpu_backward.py (not timed here)")

        def step():
            optim.zero_grad()
            net.training_step(batch, 0)["loss"].backward()
            optim.step()
        t_tr = timed(step, 3)
    except Exception as e:  # noqa: BLE001
        t_tr = str(e)[:60]
    tr = f"{t_tr * 1e3:.2f}" if isinstance(t_tr, float) else f"({t_tr})"
    print(f"| {name} | {t_inf * 1e3:.3f} | {B / t_inf:.0f} | {tr} |", flush=True)
    del net
    torch.cuda.empty_cache()
