"""Scan seeds of the three-layer gradient test (tests/test_gpu_backward.py::test_three_layer_net_gradients): per-tensor error of the HIP
backward vs float64 autograd through the oracle.  NF / SEEDS / FRAMES / BATCH / LAYERS / CONV_LAYERS / RESBLOCK from the environment; ROWS=1 prints every tensor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch
import ake_amd
import test_gpu_backward as tb

nf = int(os.environ.get("NF", 4))
for seed in range(int(os.environ.get("SEEDS", 3))):
    opt = Namespace(conv_layers=int(os.environ.get("CONV_LAYERS", 2)), n_filters=nf, head_layers=2, time_pool_size=2, genre=True, max_pool=False, frames=5,
                    resblock=bool(int(os.environ.get("RESBLOCK", 0))), pc2p_mem=bool(int(os.environ.get("PC2P_MEM", 0))),
                    p2pc_conv=bool(int(os.environ.get("P2PC_CONV", 0))), stay_sixth=bool(int(os.environ.get("STAY_SIXTH", 0))))
    torch.manual_seed(5 + seed)
    net = ake_amd.PitchClassNet(288, 12, int(os.environ.get("LAYERS", 3)), 7, opt)
    sd32 = {k: v.clone() for k, v in net.state_dict().items()}
    x, seq, labels = tb.make_case(int(os.environ.get("BATCH", 2)), int(os.environ.get("FRAMES", 96)), seed)
    loss_ref, ref = tb.reference_grads(sd32, x, seq, labels)
    net = net.cuda().train()
    out = net(x.cuda(), seq.cuda())
    loss = tb.loss_fn(out[0], out[1], out[2], *(t.cuda() for t in labels))
    loss.backward()
    rows = tb.grad_errors(net, ref)
    print(f"nf={nf} seed={seed}: loss diff {abs(float(loss.detach()) - loss_ref):.2e} worst {rows[0][0]:.2e} ({rows[0][1]}), median {rows[len(rows)//2][0]:.2e}, "
          f"tensors>1e-4: {sum(r[0] > 1e-4 for r in rows)} of {len(rows)}", flush=True)
    if os.environ.get("ROWS"):
        for e, name, m in sorted(rows, key=lambda r: r[1]):
            print(f"   {name:50s} {e:.2e}  max|ref| {m:.2e}")
