"""Count LeakyReLU kink flips between the device's f32 training forward and the f64 oracle at the last pitch-stream
BatchNorm of the default net, and list per-parameter gradient errors (see tests/test_gpu_backward.py for why these
two are linked).  Usage on a GPU box: DBG_B=4 DBG_T=52 DBG_SEED=4 python tests/tools/debug_bwd.py"""
import ctypes as C
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import numpy as np, torch
import ake_amd
from oracle import pcnet_oracle
import test_gpu_backward as tb

gold = np.load("tests/golden/pcnet_default.npz")
sd32 = {k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd/")}
opt = Namespace(**json.loads(str(gold["opt"])))
net = ake_amd.PitchClassNet(288, 12, 2, 7, opt); net.load_state_dict(sd32); net = net.cuda().train()
B, T, seed = int(os.environ.get("DBG_B", 4)), int(os.environ.get("DBG_T", 52)), int(os.environ.get("DBG_SEED", 4))
x, seq, labels = tb.make_case(B, T, seed)
sd = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd32.items()}
taps = {}
out = pcnet_oracle.pcnet_forward(sd, x.double(), seq, training=True, taps=taps)
for v in taps.values():
    if v.requires_grad: v.retain_grad()
tb.loss_fn(out[0], out[1], out[2], *labels).backward()
o = net(x.cuda(), seq.cuda())
tb.loss_fn(o[0], o[1], o[2], *(t.cuda() for t in labels)).backward()


def tap(name):
    L = ake_amd._lib.lib(); shape = (C.c_int64 * 4)()
    ake_amd._lib.check(L.ake_pcnet_tap_info(net._h, name.encode(), B, T, shape), "tap_info")
    t = torch.empty(tuple(shape), dtype=torch.float32, device="cuda")
    ake_amd._lib.check(L.ake_pcnet_tap_copy(net._h, name.encode(), B, T, net._ws_last.data_ptr(), t.data_ptr(), torch.cuda.current_stream().cuda_stream), "tap_copy")
    return t.cpu().double()


z = tap("train:z_p_last")                       # raw output of the last pitch conv (device)
bnp = "model.1.p2p.layer.7"
gamma, beta = sd32[bnp + ".weight"].double(), sd32[bnp + ".bias"].double()
mu = z.mean(dim=(0, 2, 3), keepdim=True); var = z.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
pre = (z - mu) / torch.sqrt(var + 1e-5) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
a_ref = taps["model.1.p2p.layer.8"].detach(); ga = taps["model.1.p2p.layer.8"].grad
flip = (pre > 0) != (a_ref > 0)
print("kink flips at", bnp, ":", int(flip.sum()), "of", flip.numel())
for ix in torch.nonzero(flip)[:10]:
    ix = tuple(ix.tolist())
    print("  at", ix, "pre(device)", float(pre[ix]), "act(oracle)", float(a_ref[ix]), "ga", float(ga[ix]), "-> d(dbeta) =", 0.99 * float(ga[ix]))
print("dbeta oracle", [round(v, 8) for v in sd[bnp + ".bias"].grad.tolist()])
print("dbeta device", [round(v, 8) for v in net.model[1].p2p.layer[7].bias.grad.cpu().tolist()])
for name, p in net.named_parameters():
    gr = sd[name].grad
    if p.grad is None or gr is None or float(gr.abs().max()) < 1e-9: continue
    print(f"{name:45s} max|ref| {float(gr.abs().max()):.3e}  rel err {float((p.grad.cpu().double() - gr).abs().max() / gr.abs().max()):.3e}")
