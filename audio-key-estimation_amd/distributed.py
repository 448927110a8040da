"""One process per GPU: clip sharding and result collection over torch.distributed (RCCL on ROCm).

The hot path has no cross-clip dependency (eval-mode BatchNorm uses running statistics), so inference
partitions the clip range contiguously over the ranks with NO collective in the data path; the only
communication is collecting the (clips, 35) result rows and the max-over-ranks step time.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def _forced() -> bool:
    """AKE_FORCE_PROCESS_GROUP=1: build the process group and run every collective even with ONE rank.  A one-rank RCCL communicator
    still goes through communicator init, the dtype / bucket checks, ProcessGroupNCCL's stream + event ordering and a real
    ncclAllReduce / ncclBroadcast / ncclAllGather launch -- the only way to exercise the `nccl` branches on a one-GPU box
    (tests/test_gpu_rccl.py).  A failure raises: there is no fallback to gloo or to skipping the collective."""
    return os.environ.get("AKE_FORCE_PROCESS_GROUP", "0") == "1"


def _active() -> bool:
    return dist.is_initialized() and (dist.get_world_size() > 1 or _forced())


def init_from_env(backend: str | None = None):
    """(rank, world, local_rank); initialises the default group when launched under torch.distributed.run (or, with
    AKE_FORCE_PROCESS_GROUP=1, also for a single rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or _forced()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous split; the first ``n % world`` ranks take one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_rows(local: torch.Tensor, n_total: int) -> torch.Tensor | None:
    """Concatenate each rank's (rows_r, C) block in rank order; every rank gets the (n_total, C) result."""
    if not _active():
        return local
    world = dist.get_world_size()
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], 0)


def max_over_ranks(value: float, device=None) -> float:
    if not _active():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if _active():
        dist.barrier()


# ---- data-parallel training: replicated weights, one all-reduce per optimizer step ----------------

def broadcast_parameters(net, src: int = 0):
    """Identical weights on every rank by construction: rank ``src``'s float state (one flat buffer on the device) wins."""
    if not _active():
        return
    if next(net.parameters()).device.type == "cuda":
        flat, _ = net.flat_parameters()
        if net._attached:
            if dist.get_backend() == "gloo":                    # CPU rehearsal of the N > 1 path: stage through the host
                host = flat.cpu()
                dist.broadcast(host, src)
                flat.copy_(host)
            else:
                dist.broadcast(flat, src)
            net.mark_parameters_changed()
            return
    with torch.no_grad():
        for _, v in sorted(net.state_dict(keep_vars=True).items()):
            dist.broadcast(v.data, src)


def _all_reduce_sum(t: torch.Tensor):
    if t.is_cuda and dist.get_backend() == "gloo":              # rehearsal without RCCL (several ranks on one GPU)
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def all_reduce_gradients(net) -> float:
    """SUM the gradients over the ranks and return the factor that turns the sum into the mean (1/world).

    With the parameters in the module's flat device buffer this is ONE collective on ONE 668 KB bucket (default net), called
    right after the last weight-gradient kernel was enqueued.  torch's ProcessGroupNCCL runs it on its OWN stream: it records an
    event on the caller's current stream, makes the collective stream wait for it, and (synchronous op) makes the current stream
    wait for the collective's end event before returning -- so the Adam kernel enqueued next on the compute stream sees the reduced
    buffer without any host synchronisation.  At 668 KB the ring is latency-bound, so one call beats bucketing / overlap.
    BatchNorm stays local to each rank (torch DDP's default), see DESIGN.md."""
    if not _active():
        return 1.0
    world = dist.get_world_size()
    if getattr(net, "_attached", False) and net._grads_in_place():
        _all_reduce_sum(net._flat_grad)
        return 1.0 / world
    grads = [p.grad for p in net.parameters() if p.grad is not None]
    if grads:
        flat = torch.cat([g.reshape(-1) for g in grads])
        _all_reduce_sum(flat)
        flat.mul_(1.0 / world)
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
    return 1.0
