"""N>1 host logic on CPU: 2 gloo ranks shard a clip range, run the (oracle) compute on their shard, gather."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import golden_state_dict, load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import ake_amd.distributed as D
    from oracle import pcnet_oracle
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    gold = load_golden("pcnet_default.npz")
    sd = golden_state_dict(gold, torch.float32)
    g = torch.Generator().manual_seed(0)
    x = torch.rand((n_items, 1, 288, 28), generator=g) * 2.5          # every rank builds the same clip set ...
    lo, hi = D.shard_range(n_items, r, w)                             # ... and computes only its shard
    k, t, gn = pcnet_oracle.pcnet_forward(sd, x[lo:hi], None)
    rows = D.gather_rows(torch.cat([k, t, gn], 1), n_items)
    tmax = D.max_over_ranks(1.0 + r)
    D.barrier()
    if r == 0:
        np.save(os.path.join(out_dir, "rows.npy"), rows.numpy())
        np.save(os.path.join(out_dir, "tmax.npy"), np.array(tmax))


def test_shard_range_covers_everything():
    import ake_amd.distributed as D
    for n, w in ((2048, 8), (604, 8), (5, 2), (3, 4)):
        spans = [D.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_two_rank_sharded_inference_equals_single_process(tmp_path):
    n_items = 5                                                      # uneven split: 3 + 2
    mp.spawn(_worker, args=(2, _free_port(), n_items, str(tmp_path)), nprocs=2, join=True)
    rows = np.load(tmp_path / "rows.npy")
    from oracle import pcnet_oracle
    gold = load_golden("pcnet_default.npz")
    sd = golden_state_dict(gold, torch.float32)
    g = torch.Generator().manual_seed(0)
    x = torch.rand((n_items, 1, 288, 28), generator=g) * 2.5
    k, t, gn = pcnet_oracle.pcnet_forward(sd, x, None)
    ref = torch.cat([k, t, gn], 1).numpy()
    assert rows.shape == (5, 35) and np.abs(rows - ref).max() < 1e-5
    assert float(np.load(tmp_path / "tmax.npy")) == 2.0


def _dp_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import ake_amd.distributed as D
    D.init_from_env("gloo")
    torch.manual_seed(rank)                                          # ranks start from different weights
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 2))
    D.broadcast_parameters(net)                                      # -> rank 0's weights and buffers everywhere
    g = torch.Generator().manual_seed(7)
    x = torch.randn((8, 6), generator=g)
    lo, hi = D.shard_range(8, rank, world)
    net(x[lo:hi]).square().sum().backward()
    scale = D.all_reduce_gradients(net)                              # generic (non-flat) branch: grads hold the mean afterwards
    assert scale == 1.0
    torch.save({"w": [p.detach().clone() for p in net.parameters()], "g": [p.grad.clone() for p in net.parameters()],
                "local": [p for p in net.parameters()] and None}, os.path.join(out_dir, f"dp{rank}.pt"))


def test_two_rank_gradient_all_reduce_is_the_mean(tmp_path):
    """Training partitioning of SURVEY.md section 8e on CPU: broadcast, local backward on a shard, summed all-reduce / world."""
    mp.spawn(_dp_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "dp0.pt"), torch.load(tmp_path / "dp1.pt")
    for wa, wb in zip(a["w"], b["w"]):
        assert torch.equal(wa, wb)
    for ga, gb in zip(a["g"], b["g"]):
        assert torch.equal(ga, gb)
    # emulate: same initial weights (rank 0's seed), per-shard gradients, mean
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 2))
    g = torch.Generator().manual_seed(7)
    x = torch.randn((8, 6), generator=g)
    grads = []
    for lo, hi in ((0, 4), (4, 8)):
        net.zero_grad()
        net(x[lo:hi]).square().sum().backward()
        grads.append([p.grad.clone() for p in net.parameters()])
    for got, g0, g1 in zip(a["g"], *grads):
        assert torch.allclose(got, 0.5 * (g0 + g1), atol=1e-6)
