#!/usr/bin/env python3
"""BASELINE configs[4]: equivariance_test.py on the HIP path, the shifted inputs split over the ranks.

The reference script (equivariance_test.py:172-205) pads a guard octave above and below a 288-bin CQT (-> 360 bins), shifts it
by +-1..12 semitones (zero fill) and stacks the 25 key outputs.  Here the 25 inputs are dealt to the ranks (one process per
GPU, no collective in the data path), every rank runs its share through the 360-bin net of the reference-generated fixture,
the (25, 12) tables are gathered, and rank 0 asserts (a) the circular-shift identity of SURVEY 4.2 on key and tonic and
(b) equality with the reference's own table.  One JSON line.

    python3 tools/config5_equivariance.py                       # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/config5_equivariance.py
    AKE_REHEARSE_ONE_GPU=1 ... --nproc-per-node 2 ...           # functional rehearsal: every rank on cuda:0, gloo collectives
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import numpy as np
import torch
import ake_amd
from ake_amd import distributed as D

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shift_up(mel, s):        # equivariance_test.py:122-133: +s semitones = rows move up by 3s bins, zero fill below
    out = np.zeros_like(mel)
    if s == 0:
        return mel.copy()
    out[3 * s:] = mel[:-3 * s]
    return out


def shift_down(mel, s):      # equivariance_test.py:135-146
    out = np.zeros_like(mel)
    out[:-3 * s] = mel[3 * s:]
    return out


def main():
    if os.environ.get("AKE_REHEARSE_ONE_GPU"):
        rank, world, _ = D.init_from_env("gloo"); local_rank = 0
    else:
        rank, world, local_rank = D.init_from_env()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    gold = np.load(os.path.join(REPO, "tests", "golden", "pcnet_guard360.npz"))
    opt = Namespace(**json.loads(str(gold["opt"])))
    sd = {k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd/")}
    net = ake_amd.PitchClassNet(opt.octaves * 36, 12, opt.num_layers, opt.kernel_size, opt)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    mel = gold["mel"].astype(np.float64)
    T = mel.shape[1]
    mel_g = np.concatenate([np.zeros((36, T)), mel, np.zeros((36, T))])
    # row order of the reference's table: +12 ... +1, 0, -1 ... -12 (equivariance_test.py:183-198)
    inputs = [shift_up(mel_g, s) for s in range(12, 0, -1)] + [mel_g] + [shift_down(mel_g, s) for s in range(1, 13)]
    lo, hi = D.shard_range(len(inputs), rank, world)
    seq = torch.tensor(T).reshape(1, 1)
    D.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
    rows = []
    for x in inputs[lo:hi]:                                           # B = 1 per forward, as the script runs it
        k, t = net(torch.from_numpy(x).reshape(1, 1, 360, T).to(dev), seq)
        rows.append(torch.cat([k[0], t[0]]).float())
    local = torch.stack(rows) if rows else torch.zeros((0, 24), device=dev)
    torch.cuda.synchronize(); D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0, None if os.environ.get("AKE_REHEARSE_ONE_GPU") else dev)
    table = D.gather_rows(local.cpu() if os.environ.get("AKE_REHEARSE_ONE_GPU") else local, len(inputs))
    if rank != 0:
        return
    table = table.cpu().double().numpy()
    K, Tn = table[:, :12], table[:, 12:]
    roll_err = 0.0
    for s in range(1, 13):
        roll_err = max(roll_err, np.abs(K[12 - s] - np.roll(K[12], s)).max(), np.abs(K[12 + s] - np.roll(K[12], -s)).max(),
                       np.abs(Tn[12 - s] - np.roll(Tn[12], s)).max(), np.abs(Tn[12 + s] - np.roll(Tn[12], -s)).max())
    ref_err = max(np.abs(K - gold["key_eval"]).max() / np.abs(gold["key_eval"]).max(), np.abs(Tn - gold["tonic_eval"]).max() / np.abs(gold["tonic_eval"]).max())
    ok = bool(roll_err <= 1e-5 and ref_err < 1e-4)
    print(json.dumps({"config": "BASELINE configs[4]: 25 shifted 360-bin inputs, eval mode, sharded over the ranks", "n_gpus": world,
                      "inputs": len(inputs), "max_roll_identity_error": float(roll_err), "max_rel_error_vs_reference_table": float(ref_err),
                      "seconds": round(dt, 4), "pass": ok}))
    assert ok, (roll_err, ref_err)


if __name__ == "__main__":
    main()
