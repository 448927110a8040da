"""RCCL for real on the one GPU of the test box: a world_size-1 `nccl` process group in a FRESH child process (the
group is initialised before that process makes any other GPU call), through which run

  * `broadcast_parameters`   (ncclBroadcast of the module's flat 668 KB device buffer),
  * `all_reduce_gradients`   (ncclAllReduce SUM on the flat gradient buffer, then the fused Adam with grad_scale = 1/world),
  * `gather_rows`            (ncclAllGather of the (clips, 35) result rows),
  * `max_over_ranks`, `barrier`,
  * one `Trainer.fit` over three batches (each optimizer step all-reduces the gradient bucket),

and whose weights must equal, bit for bit, those of the same three steps without any process group (a one-rank SUM is the identity, so
any difference is a stream-ordering or dtype bug of the collective path).  `AKE_FORCE_PROCESS_GROUP=1` is what makes
`ake_amd.distributed` build the group and issue the collectives with a single rank; an RCCL error exits non-zero -- there is no gloo
fallback in the child.  N > 1 over xGMI stays unmeasured on this one-GPU box (README)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, os.environ["AKE_REPO"]); sys.path.insert(0, os.path.join(os.environ["AKE_REPO"], "tests"))
import torch
import torch.distributed as dist
import ake_amd
import ake_amd.distributed as D
rank, world, local_rank = D.init_from_env("nccl")              # RCCL; raises on failure
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1 and D._active()
from test_gpu_training import default_net, make_batch
from conftest import load_golden
DEV = "cuda:0"
gold = load_golden("pcnet_default.npz")

def three_steps(collectives):
    net, _ = default_net(gold)
    net = net.to(DEV)
    batches = [make_batch(6, 40, 200 + i) for i in range(3)]
    if collectives:
        trainer = ake_amd.Trainer(max_epochs=1, accumulate_grad_batches=1)
        trainer.fit(net, train_dataloaders=batches)            # broadcast_parameters + all_reduce_gradients per step
    else:
        optim = net.configure_optimizers()[0][0]
        net.train()
        for i, b in enumerate(batches):
            optim.zero_grad()
            net.training_step({k: v.to(DEV) for k, v in b.items()}, i)["loss"].backward()
            optim.step()
    return net

calls = {"all_reduce": 0, "broadcast": 0, "all_gather": 0}
for name in calls:
    orig = getattr(dist, name)
    def counted(*a, _o=orig, _n=name, **k):
        calls[_n] += 1
        return _o(*a, **k)
    setattr(dist, name, counted)

net = three_steps(True)
flat_pg = net.flat_parameters()[0].clone()
assert calls["broadcast"] == 1 and calls["all_reduce"] == 3, calls
# the collectives one by one on the real buffers
g = net._flat_grad
net.zero_grad()
net.train()
net.training_step({k: v.to(DEV) for k, v in make_batch(6, 40, 7).items()}, 0)["loss"].backward()
before = g.clone()
assert g.is_cuda and g.dtype == torch.float32 and g.numel() >= 167031
assert D.all_reduce_gradients(net) == 1.0                       # 1 / world
torch.cuda.synchronize()
assert torch.equal(g, before) and float(g.abs().max()) > 0      # SUM over one rank
rows = torch.arange(7 * 35, dtype=torch.float32, device=DEV).reshape(7, 35)
got = D.gather_rows(rows, 7)
assert calls["all_gather"] == 1 and torch.equal(got, rows)
assert D.max_over_ranks(1.25, torch.device(DEV)) == 1.25
D.barrier()
os.environ["AKE_FORCE_PROCESS_GROUP"] = "0"                     # the same three steps without any collective
assert not D._active()
ref = three_steps(False).flat_parameters()[0]
assert torch.equal(flat_pg, ref), float((flat_pg - ref).abs().max())
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK", calls)
"""


def test_one_rank_rccl_process_group(tmp_path):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, AKE_REPO=REPO, AKE_FORCE_PROCESS_GROUP="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = tmp_path / "rccl_child.py"
    script.write_text(CHILD)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-6000:])
    print(r.stderr[-6000:])
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout


def test_bench_train_line_through_a_one_rank_rccl_group(tmp_path):
    """`bench.py --train` with the forced group: the line's step really contains the ncclAllReduce of the gradient bucket (and carries the
    step-0 parity object).  Small batch: a functional run, not a measurement."""
    import json
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, AKE_FORCE_PROCESS_GROUP="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--train", "--batch", "16", "--steps", "3", "--warmup", "1",
                        "--spinup-seconds", "0"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["collective_backend"] == "nccl" and line["parity"]["max_rel_err"] < 2e-4
