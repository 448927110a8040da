#!/usr/bin/env python3
"""Headline benchmark: clips/s (15 s @ 22.05 kHz) through HIP CQT + PitchClassNet forward.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

One "step" = one pass of the hot path (``ake_pipeline_forward_f32``: CQT -> seq_length fill ->
network) over B = 256 synthetic sine-mix clips that are already resident in HBM (BASELINE.json
configs[1]).  For N > 1 the driver launches this file under ``torch.distributed.run``; every rank
owns 256 clips of its own (weak scaling, clips are independent, no data-path collective) and the
step time is the max over ranks.  Rank 0 prints ONE JSON line.

Extra objects on the line (tier contract):
  roofline      dominant kernel (conv_p2p_f16_kernel: the 7x7 circular pitch convolutions, 65 % of
                the MACs): algorithmic FLOPs (2 x MACs) of its launches / their hipEvent-measured
                duration, vs the 2.5 PFLOP/s dense f16 / bf16 MFMA peak of MI355X; the kernel multiplies
                f16 activations with f16 weights, one MFMA product per MAC.
  roofline_cqt  the CQT stage against the 8 TB/s HBM peak (1 410 552 algorithmic bytes per clip).
  cpu_baseline  the CPU oracle (direct-form CQT as BLAS matmuls + the oracle network) timed on this
                box's host cores on a bounded sample: BASELINE.md section 3's cases (fp32 / fp64, B = 1 /
                B = 32, CQT and network separately, best of several thread counts); `value` = the
                float64 (reference dtype) B = 1 (eval.py's batch size) whole-path rate.
  parity        max relative error of four clips of the LAST timed step against the CPU oracle
                (outside the timed region) and a MIREX sanity score of the whole batch.
  sustained     >= 2 s of steps rotating over >= 4 distinct resident batches (> 256 MiB of audio, so
                no step is served from the Infinity Cache), with its own ms_per_step and the dominant
                kernel's mean launch duration at that steady state.

`--gpus N` without a launcher (WORLD_SIZE unset) starts N fresh rank processes itself
(torch.distributed.run, one per GPU, RCCL) BEFORE this process touches the GPU and relays their
output; a world size that differs from --gpus is a hard error.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from argparse import Namespace

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

SR, FRAMES, OCTAVES = 22050, 5, 8
N_SAMPLES = SR * 15                       # 330 750
HOP = 4410
T_FRAMES = 1 + N_SAMPLES // HOP           # 76
P = 36 * OCTAVES
# SURVEY.md section 8d / DESIGN.md "Measurement"
CQT_BYTES_PER_CLIP = N_SAMPLES * 4 + P * T_FRAMES * 4                  # 1 410 552
P2P_MACS_PER_CLIP = (5 * 8 + 8 * 8 + 8 * 8) * 49 * P * T_FRAMES        # 180 182 016 (three 7x7 convs: 5->8, 8->8, 8->8)
assert P2P_MACS_PER_CLIP == 180_182_016
PEAK_BF16_TFLOPS = 2500.0
NET_MACS_PER_CLIP = 277_395_712
PEAK_FP32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def kernel_set_hash():
    """sha256 over the kernel sources (csrc/*.hip, *.h, *.cpp, build.sh): identifies the kernel set a profile was taken with."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(REPO, "audio-key-estimation_amd", "csrc")
    for p in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.cpp"))
                    + [os.path.join(csrc, "build.sh")]):
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 counter passes (tools/pmc_traffic.py; FETCH_SIZE x2 on gfx950, see
    DESIGN.md "Measurement").  bench.py cannot collect PMC counters itself (they need the profiler around the process), so the
    line carries the numbers of the newest profile in profiles/ that was taken with this same command and batch size -- and ONLY
    if that profile was taken with the kernel sources as they are now (`kernel_set` hash stored by tools/pmc_traffic.py);
    otherwise -> (name, {}, reason): `traffic` is null and the line says why."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, {}, "no profiles/r*_pmc_traffic.json"
    with open(files[-1]) as f:
        d = json.load(f)
    name, have, want = os.path.basename(files[-1]), d.get("kernel_set"), kernel_set_hash()
    if have != want:
        why = f"STALE: profiles/{name} was taken with kernel set {have}, the sources now hash to {want}; re-run tools/collect_round.sh"
        print("bench.py: PMC traffic " + why, file=sys.stderr)
        return name, {}, why
    return name, d["kernels"], None


def load_fixture_weights():
    gold = np.load(os.path.join(REPO, "tests", "golden", "pcnet_default.npz"))
    return {k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd/")}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def _median_time(fn, budget_s, min_iters=1, max_iters=20):
    """Median wall time of fn() over as many runs as fit the budget (one untimed warm-up first)."""
    fn()
    ts, t_end = [], time.perf_counter() + budget_s
    while len(ts) < max_iters and (len(ts) < min_iters or time.perf_counter() < t_end):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), len(ts)


def cpu_baseline(sd, budget_s=20.0):
    """Oracle ("port") on the host cores, BASELINE.md section 3: the same synthetic clips; float32 and float64 (the reference's
    dtype); B = 1 (eval.py's batch size, BASELINE configs[0]) and B = 32; CQT and network timed separately; best of the thread
    counts {8, 32, all}.  Bounded: every (case, thread count) gets an equal share of `budget_s`."""
    from ake_amd import synthetic
    from oracle import cqt_oracle, pcnet_oracle
    ncpu = os.cpu_count() or 1
    prev = torch.get_num_threads()
    thread_opts = sorted({min(8, ncpu), min(32, ncpu), ncpu})
    audio, _ = synthetic.make_batch(range(32))
    cases, share = [], budget_s / (2 * 2 * 2 * len(thread_opts))
    try:
        for dtype, dname in ((torch.float32, "f32"), (torch.float64, "f64")):
            cqt = cqt_oracle.FastDirectCQT(SR, HOP, dtype=dtype)
            sdd = pcnet_oracle.to_dtype(sd, dtype)
            for B in (1, 32):
                y = torch.from_numpy(audio[:B])
                seq = torch.full((B,), T_FRAMES)
                with torch.no_grad():
                    mel = cqt(y)[:, None].contiguous()
                    best = {}
                    for stage, fn in (("cqt", lambda: cqt(y)), ("net", lambda: pcnet_oracle.pcnet_forward(sdd, mel, seq))):
                        for th in thread_opts:
                            torch.set_num_threads(th)
                            t, n = _median_time(fn, share)
                            if stage not in best or t < best[stage][0]:
                                best[stage] = (t, th, n)
                t_all = best["cqt"][0] + best["net"][0]
                cases.append({"dtype": dname, "batch": B,
                              "cqt_ms": round(best["cqt"][0] * 1e3, 2), "cqt_threads": best["cqt"][1],
                              "net_ms": round(best["net"][0] * 1e3, 2), "net_threads": best["net"][1],
                              "iters": [best["cqt"][2], best["net"][2]],
                              "clips_per_s": round(B / t_all, 2), "cqt_clips_per_s": round(B / best["cqt"][0], 2),
                              "net_clips_per_s": round(B / best["net"][0], 2)})
    finally:
        torch.set_num_threads(prev)
    ref = next(c for c in cases if c["dtype"] == "f64" and c["batch"] == 1)
    top = max(cases, key=lambda c: c["clips_per_s"])
    return {"value": ref["clips_per_s"], "unit": "clips/s", "cores": max(ref["cqt_threads"], ref["net_threads"]), "kind": "port",
            "sample": f"oracle direct-form CQT (BLAS matmuls) + oracle PitchClassNet forward (torch CPU ops) on 1 and 32 synthetic 15 s clips; "
                      f"value = float64 (the reference's dtype), batch 1 (eval.py), median of {ref['iters']} runs (CQT, net), best of "
                      f"{thread_opts} threads per stage; every case in `cases`",
            "cpu_model": _cpu_model(), "host_cores": ncpu, "thread_counts_tried": thread_opts,
            "best": {"value": top["clips_per_s"], "dtype": top["dtype"], "batch": top["batch"]},
            "cases": cases}


def launch_ranks(n):
    """--gpus N without a launcher: start N fresh rank processes (one per GPU, RCCL) from THIS process, which has made no GPU
    call, relay their output, and exit with their code.  The ranks re-enter this file with WORLD_SIZE set."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        print(f"bench.py: the {n}-rank launch failed with exit code {rc} (no single-rank or gloo fallback)", file=sys.stderr)
    sys.exit(rc)


def require_world(world, gpus):
    if world != gpus:
        sys.exit(f"bench.py: WORLD_SIZE={world} but --gpus {gpus}: refusing to print a line for a world size that was not asked for")


def init_ranks(D):
    """(rank, world, local_rank).  AKE_REHEARSE_ONE_GPU=1 runs the N > 1 code path on a one-GPU box: every rank uses cuda:0
    and the collectives go through gloo (RCCL needs one GPU per rank) -- a functional rehearsal, never a measurement."""
    if os.environ.get("AKE_REHEARSE_ONE_GPU"):
        rank, world, _ = D.init_from_env("gloo")
        return rank, world, 0
    return D.init_from_env("nccl")                                     # RCCL; an init failure raises (non-zero exit), never gloo


def train_main(args):
    """BASELINE configs[3]: B clips per GPU (log-CQT features resident, as the reference caches them, KeyDataset.py:154-192),
    one training step = train-mode forward + general_step loss + HIP backward + ONE all-reduce of the flat 668 KB gradient
    buffer + fused Adam.  Secondary line; the headline metric is the inference line printed without --train."""
    import ake_amd
    from ake_amd import distributed as D, synthetic
    rank, world, local_rank = init_ranks(D)
    require_world(world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sd = load_fixture_weights()
    net = ake_amd.PitchClassNet(P, 12, 2, 7, Namespace(genre=True, lr=3e-4, gamma=0.96, acc_grad=1))
    net.load_state_dict(sd, strict=True)
    net = net.to(dev)
    B = args.batch
    audio, labels = synthetic.make_batch_device(range(rank * B, rank * B + B), dev)
    mel = ake_amd.cqt_logmag(audio, SR, HOP, n_bins=P, bins_per_octave=36)[:, None].contiguous()   # (B,1,288,76), computed once
    del audio
    batch = {"mel": mel, "seq_length": torch.full((B,), T_FRAMES, device=dev), **{k: torch.as_tensor(v).to(dev) for k, v in labels.items()}}
    D.broadcast_parameters(net)
    optim = net.configure_optimizers()[0][0]
    net.train()
    net.trainer = ake_amd.Trainer(accumulate_grad_batches=1)
    parity = None
    if not args.no_parity:
        parity = train_parity(net, sd, batch, rank)                     # step 0, outside the timed region (weights untouched: no optimizer step)

    def step(i):
        optim.zero_grad()
        net.training_step(batch, i)["loss"].backward()
        optim.grad_scale = D.all_reduce_gradients(net)
        optim.step()

    # device / host spin-up as in the inference line (--spinup-seconds; untimed, not part of W): the first ~50 steps of a fresh process run
    # 20-60 % slower than the steady state (measured: 20 timed steps after 5 warm-up steps 11.0-15.1 ms, 150 after 30: 9.12 +- 0.01 ms)
    # (a step holds a collective: every rank runs the same NUMBER of spin-up steps, 100 per 0.5 s asked for)
    n_spin = int(round(200 * args.spinup_seconds))
    for i in range(n_spin):
        step(i)
    for i in range(max(args.warmup, 1)):
        step(i)
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0, dev)
    ake_amd._lib.prof_enable("", True)
    for i in range(args.steps):
        step(i)
    torch.cuda.synchronize()
    prof_all = ake_amd._lib.prof_results()
    ake_amd._lib.prof_enable("", False)
    if rank != 0:
        return
    value = B * world * args.steps / dt
    kernel_ms = {k: round(v[0] / args.steps, 4) for k, v in sorted(prof_all.items(), key=lambda kv: -kv[1][0])}
    print(json.dumps({
        "metric": "clips/s, PitchClassNet training step (fwd + bwd + grad all-reduce + Adam)", "value": round(value, 1), "unit": "clips/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 storage, accumulation and gradients; pitch convs (forward, data gradient) as f16 hi + lo x 3 MFMA products (2^-22 of a product "
                 "dropped), their weight gradient as split bf16 x 3, everything else f32 MFMA", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[3]: {B} clips per GPU (288x76 log-CQT resident), default PitchClassNet, local BatchNorm, "
                               f"one all-reduce of the flat gradient buffer per step, fused Adam lr 3e-4",
                   "clips_per_gpu": B, "parallelism": f"data-parallel x{world}", "spinup_steps": n_spin},
        # fwd + data gradient + weight gradient = 3 x the forward's MACs, algorithmic, against the dense f16 / bf16 MFMA peak the kernels
        # multiply on (NOT the 157 TF f32 peak: VERDICT r2 item 7); every algorithmic MAC costs 3 MFMA products in these kernel families
        # (f16 hi + lo x 3, split bf16 x 3), so the matrix pipe's own ceiling for this formulation is a third of the peak
        "train_algorithmic_tflops": round(3 * 2.0 * NET_MACS_PER_CLIP * value / world / 1e12, 2),
        "train_frac_of_mfma_peak": round(3 * 2.0 * NET_MACS_PER_CLIP * value / world / (PEAK_BF16_TFLOPS * 1e12), 4),
        "mfma_products_per_mac": 3,
        "collective_backend": (torch.distributed.get_backend() if torch.distributed.is_initialized() else None),
        "parity": parity,
        "kernel_ms_per_step": kernel_ms}))


def train_parity(net, sd, batch, rank):
    """Step-0 check of the training line, outside the timed region: the train-mode forward's three outputs and the
    general_step loss of THIS rank-0 batch against the float64 oracle forward (batch statistics) + the numpy loss restatement, and a
    gradient sanity (finite, non-zero, the same bits when the step is repeated).  Asserted: outputs 2e-4, loss 2e-5."""
    from oracle import loss_oracle, pcnet_oracle
    for p in net.parameters():
        p.grad = None
    d = net.training_step(batch, 0)
    d["loss"].backward()
    g1 = torch.cat([p.grad.detach().flatten() for p in net.parameters()]).clone()
    for p in net.parameters():
        p.grad = None
    net.training_step(batch, 0)["loss"].backward()
    g2 = torch.cat([p.grad.detach().flatten() for p in net.parameters()])
    with torch.no_grad():
        out = net(batch["mel"], batch["seq_length"])
    torch.cuda.synchronize()
    assert bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0, "training step: gradients are not finite / all zero"
    assert torch.equal(g1, g2), "training step: two backward passes of the same batch differ (the step must be bit-reproducible)"
    if rank != 0:
        return None
    t0 = time.perf_counter()
    sd64 = pcnet_oracle.to_dtype(sd, torch.float64)
    prev = torch.get_num_threads()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))       # (a GPU box grants 16 cores whatever it reports)
    with torch.no_grad():
        ref = pcnet_oracle.pcnet_forward(sd64, batch["mel"].double().cpu(), batch["seq_length"].cpu(), training=True)
    torch.set_num_threads(prev)
    loss_ref = loss_oracle.general_step_loss(ref[0].numpy(), ref[1].numpy(), ref[2].numpy(), batch["key_labels"].cpu().numpy(),
                                             batch["tonic_labels"].cpu().numpy(), batch["genre"].cpu().numpy())
    errs = {n: float((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-6)) for n, a, b in zip(("key", "tonic", "genre"), out, ref)}
    loss = float(d["loss"].detach())
    res = {"max_rel_err": round(max(errs.values()), 8), "per_output": {k: round(v, 8) for k, v in errs.items()}, "tolerance": 2e-4,
           "loss": round(loss, 7), "loss_oracle": round(loss_ref, 7), "loss_rel_err": round(abs(loss - loss_ref) / max(1.0, abs(loss_ref)), 9),
           "grad_abs_max": float(g1.abs().max()), "backward_bit_reproducible": True, "clips": int(batch["mel"].shape[0]),
           "oracle_seconds": round(time.perf_counter() - t0, 1),
           "against": "float64 oracle forward in train mode (batch statistics; pinned on the reference) + numpy restatement of general_step's loss, "
                      "the whole rank-0 batch at step 0, outside the timed region; gradients at this batch size vs float64 autograd: "
                      "tests/test_gpu_train_scale.py"}
    assert res["max_rel_err"] < 2e-4 and res["loss_rel_err"] < 2e-5, f"training parity vs the oracle: {res}"
    return res


def parity_and_mirex(rows, audio_last, first_clip, world, B):
    """Outside the timed region: four clips of the last timed step (rank 0's shard: first, last and two inner M-tile boundaries of the
    CQT bank's 16-clip tiles) against the CPU oracle (float64 direct-form CQT -> float64 network), and a MIREX sanity score of
    every rank's rows against the synthetic labels (seeded fixture weights: a sanity signal, not a quality claim)."""
    import ake_amd
    from ake_amd import synthetic
    from oracle import cqt_oracle, pcnet_oracle
    sd64 = pcnet_oracle.to_dtype(load_fixture_weights(), torch.float64)
    pick = sorted({0, min(B - 1, 15), min(B - 1, B // 2 + 16), B - 1})
    y = audio_last[pick].cpu()
    with torch.no_grad():
        mel = cqt_oracle.FastDirectCQT(SR, HOP, dtype=torch.float64)(y)
        ref = torch.cat(pcnet_oracle.pcnet_forward(sd64, mel[:, None], torch.full((len(pick),), T_FRAMES)), 1)
    got = rows[pick].double().cpu()
    errs = {}
    for name, sl in (("key", slice(0, 12)), ("tonic", slice(12, 24)), ("genre", slice(24, 35))):
        errs[name] = float((got[:, sl] - ref[:, sl]).abs().max() / ref[:, sl].abs().max().clamp_min(1e-6))
    labs = [synthetic.clip_recipe(first_clip(r) + i)[4] for r in range(world) for i in range(B)]
    L = {k: torch.from_numpy(np.stack([l[k] for l in labs])) for k in labs[0]}
    r = rows.float().cpu()
    m = ake_amd.metrics.mirex_score(L["key_labels"], r[:, :12], L["tonic_labels"], r[:, 12:24], L["key_signature_id"])
    return ({"max_rel_err": round(max(errs.values()), 8), "per_output": {k: round(v, 8) for k, v in errs.items()}, "tolerance": 1e-3,
             "clips_checked": [int(i) for i in pick],
             "against": "oracle: float64 direct-form CQT (build's own spec, unpinned vs librosa) -> float64 PitchClassNet restatement (pinned "
                        "on the reference); max |a-b| / max |b| per output tensor, last timed step, outside the timed region"},
            {"score": round(float(m[0]), 4), "correct": round(float(m[1]), 4), "fifths": round(float(m[2]), 4), "relative": round(float(m[3]), 4),
             "parallel": round(float(m[4]), 4), "other": round(float(m[5]), 4), "clips": int(r.shape[0]),
             "note": "sanity only: seeded random fixture weights on synthetic labelled sine-mix clips (models.py:1065-1116 scoring)"})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU per step")
    ap.add_argument("--rotate", type=int, default=4,
                    help="distinct resident audio batches the steps rotate over (4 x 339 MB > the 256 MiB Infinity Cache: no step re-reads "
                         "audio that is still on chip)")
    ap.add_argument("--sustained-seconds", type=float, default=2.0,
                    help="length of the extra 'sustained' pass (same steps, same rotation, long enough for DVFS steady state); 0: skip")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent steps in flight per GPU in the TIMED region: steps are issued round-robin on this many streams (own "
                         "workspace each), so one batch's CQT overlaps another's convolutions; 1 (default) = strictly one step after the "
                         "other, which keeps the per-kernel durations of the roofline clean")
    ap.add_argument("--pipelined-streams", type=int, default=2,
                    help="an extra, separately timed pass of the same steps with this many in flight, reported as 'pipelined' (0: skip)")
    ap.add_argument("--spinup-seconds", type=float, default=0.5,
                    help="untimed steps issued for this long BEFORE the W warm-up steps: the first ~0.2 s after idle run 5 %% slower (clock ramp, "
                         "first touch of the workspaces; measured: W = 5 -> 0.697, W = 50 -> 0.672, W = 300 -> 0.664 ms per step), and a serving "
                         "process is never in that state; reported on the line as config.spinup_seconds, cross-checked by 'sustained'")
    ap.add_argument("--precision", choices=("mixed", "f32x3"), default="mixed",
                    help="arithmetic of the inference convolutions (ake_pcnet_config.precision): 'mixed' = f16 single-product pitch / semitone / layer-0 "
                         "convolutions + split-bf16 x 3 elsewhere (default, the headline); 'f32x3' = no operand rounded below 2^-17")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle comparison of the last step's rows (profiler runs)")
    ap.add_argument("--train", action="store_true",
                    help="BASELINE configs[3] instead of the headline: one fwd + bwd + gradient all-reduce + fused Adam step per GPU batch")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)                                         # never returns; nothing here has touched the GPU yet
    if args.train:
        return train_main(args)

    import ake_amd
    from ake_amd import distributed as D, synthetic

    rank, world, local_rank = init_ranks(D)
    require_world(world, args.gpus)
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the product path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    sd = load_fixture_weights()
    net = ake_amd.PitchClassNet(P, 12, 2, 7, Namespace(genre=True, precision=args.precision))
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    net_dtype = net.precision_dtype()                                   # read back from the device handle (ake_pcnet_precision)
    # the dominant kernel of each mode: the three 7x7 pitch convolutions (65 % of the MACs)
    # f32x3: the same persistent pitch-conv kernel with f16 hi + lo operands, three MFMA products per MAC (f32-equivalent to 2^-22)
    DOM = "conv_p2p_f16_kernel" if args.precision == "mixed" else "conv_p2p_f16x3_kernel/p2p"
    DOM_PEAK = PEAK_BF16_TFLOPS
    est = ake_amd.KeyEstimator(net, SR, FRAMES, streams=args.streams)
    est1 = est if args.streams == 1 else ake_amd.KeyEstimator(net, SR, FRAMES)      # second, untimed pass: one step after the other
    B, R = args.batch, max(1, args.rotate)
    first_clip = lambda r, j=0: (r * R + j) * B                         # each rank synthesises R batches of its own clips
    batches = [synthetic.make_batch_device(range(first_clip(rank, j), first_clip(rank, j) + B), dev)[0] for j in range(R)]
    assert all(a.shape == (B, N_SAMPLES) for a in batches)

    if args.spinup_seconds > 0:                                         # device spin-up (see --spinup-seconds); not part of W, not timed
        t_spin, i = time.perf_counter(), 0
        while time.perf_counter() - t_spin < args.spinup_seconds:
            for _ in range(16):
                est(batches[i % R])
                i += 1
            est.join()
            torch.cuda.synchronize()
    for i in range(max(args.warmup, 1)):
        out = est(batches[i % R])
        if est1 is not est:
            est1(batches[i % R])
    est.join()
    torch.cuda.synchronize()
    # timed region: hipEvents (on the launch stream) bracket only the dominant kernel -- 3 launches per step
    ake_amd._lib.lib().ake_prof_reset()
    ake_amd._lib.prof_enable(DOM, True)
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = est(batches[i % R])
    est.join()
    torch.cuda.synchronize()
    D.barrier()
    dt = time.perf_counter() - t0
    prof = ake_amd._lib.prof_results()
    last_j = (args.steps - 1) % R
    # second, untimed pass, one step after the other on one stream with every kernel bracketed: per-kernel breakdown, the CQT stage's
    # roofline, and the dominant kernel's duration when nothing else shares the GPU with it
    ake_amd._lib.prof_enable("", True)
    for i in range(args.steps):
        est1(batches[i % R])
    torch.cuda.synchronize()
    prof_all = ake_amd._lib.prof_results()
    ake_amd._lib.prof_enable("", False)
    dt = D.max_over_ranks(dt, dev)
    # third pass (timers off): the same steps with several in flight -- the throughput a serving loop over independent batches gets
    dt_pipe = None
    if args.pipelined_streams > 1 and args.streams == 1:
        estp = ake_amd.KeyEstimator(net, SR, FRAMES, streams=args.pipelined_streams)
        for i in range(2 * args.pipelined_streams):
            estp(batches[i % R])
        estp.join()
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            estp(batches[i % R])
        estp.join()
        torch.cuda.synchronize()
        D.barrier()
        dt_pipe = D.max_over_ranks(time.perf_counter() - t0, dev)
    # sustained pass: the timed region's steps again, for >= --sustained-seconds (the step count is fixed from the timed region's
    # rate so that every rank runs the same number), dominant kernel bracketed as in the timed region
    sustained = None
    if args.sustained_seconds > 0:
        n_sus = max(args.steps, int(args.sustained_seconds / (dt / args.steps) * 1.1) + 1)
        ake_amd._lib.lib().ake_prof_reset()
        ake_amd._lib.prof_enable(DOM, True)
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_sus):
            est1(batches[i % R])
        torch.cuda.synchronize()
        D.barrier()
        dt_sus = D.max_over_ranks(time.perf_counter() - t0, dev)
        prof_sus = ake_amd._lib.prof_results()
        ake_amd._lib.prof_enable("", False)
        s_ms, s_n = prof_sus.get(DOM, (0.0, 0))
        s_tf = 2.0 * P2P_MACS_PER_CLIP * B * n_sus / (s_ms * 1e-3) / 1e12 if s_ms > 0 else None
        sustained = {"seconds": round(dt_sus, 3), "steps": n_sus, "distinct_batches": R, "resident_audio_bytes": R * B * N_SAMPLES * 4,
                     "value": round(B * world * n_sus / dt_sus, 1), "unit": "clips/s", "ms_per_step": round(dt_sus / n_sus * 1e3, 4),
                     "dominant_kernel_avg_launch_ms": round(s_ms / s_n, 4) if s_n else None,
                     "dominant_kernel_tflops": round(s_tf, 2) if s_tf else None,
                     "dominant_kernel_frac": round(s_tf / DOM_PEAK, 4) if s_tf else None}

    # result collection (outside the timed region): 35 floats per clip, rank order
    rows = D.gather_rows(torch.cat(out, 1), B * world)
    if rank != 0:
        return
    assert rows.shape == (B * world, 35) and bool(torch.isfinite(rows).all())
    parity = mirex = None
    if not args.no_parity:
        parity, mirex = parity_and_mirex(rows, batches[last_j], lambda r: first_clip(r, last_j), world, B)
        assert parity["max_rel_err"] < 1e-3, f"parity vs the oracle: {parity}"

    clips = B * world * args.steps
    value = clips / dt
    # dominant kernel: the three 7x7 pitch convolutions (65 % of the network's MACs), conv_p2p_f16_kernel
    p2p_ms, p2p_n = prof.get(DOM, (0.0, 0))
    launches_per_step = p2p_n / args.steps if args.steps else 0
    p2p_flops = 2.0 * P2P_MACS_PER_CLIP * B * args.steps              # algorithmic: (5*8 + 8*8 + 8*8) * 49 MACs per position
    achieved = p2p_flops / (p2p_ms * 1e-3) / 1e12 if p2p_ms > 0 else None
    cqt_ms = sum(prof_all.get(k, (0.0, 0))[0] for k in prof_all if k.startswith("cqt_"))
    cqt_gbs = CQT_BYTES_PER_CLIP * B * args.steps / (cqt_ms * 1e-3) / 1e9 if cqt_ms > 0 else None
    kernel_ms = {k: round(v[0] / args.steps, 4) for k, v in sorted(prof_all.items(), key=lambda kv: -kv[1][0])}
    p2p1_ms, p2p1_n = prof_all.get(DOM, (0.0, 0))
    achieved1 = p2p_flops / (p2p1_ms * 1e-3) / 1e12 if p2p1_ms > 0 else None
    traffic_src, traffic, traffic_stale = pmc_traffic()
    p2p_traffic = cqt_traffic = None
    if B == 256 and traffic:
        bf_rows = [v for k, v in traffic.items() if k.startswith("conv_p2p_f16")]
        p2p_traffic = round(sum(v["hbm_bytes"] * v["dispatches"] for v in bf_rows) / max(1, sum(v["dispatches"] for v in bf_rows))) if bf_rows else None
        cqt_traffic = sum(v["hbm_bytes"] for k, v in traffic.items() if k.startswith("cqt_")) or None
    line = {
        "metric": "clips/s (15 s @ 22.05 kHz), HIP CQT + PitchClassNet forward",
        "value": round(value, 1), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": net_dtype + "; CQT filter bank: 3-term split-bf16 on MFMA, half-band decimators exact f32", "data": "synthetic",
        "precision": args.precision,
        "config": {"workload": f"BASELINE configs[1]: batch={B} synthetic 15 s sine-mix clips per GPU, HIP CQT (288 bins, hop 4410) "
                               f"+ default PitchClassNet inference (genre head on), audio resident in HBM",
                   "clips_per_gpu": B, "n_samples": N_SAMPLES, "frames": T_FRAMES, "weights": "tests/golden/pcnet_default.npz (seeded)",
                   "distinct_batches": R, "resident_audio_bytes_per_gpu": R * B * N_SAMPLES * 4, "spinup_seconds": args.spinup_seconds,
                   "parallelism": f"clip-sharded x{world}, no data-path collective",
                   "streams": args.streams,
                   "steps_in_flight": f"{args.streams}: every step is the whole path over one batch; consecutive steps go round-robin to "
                                      f"{args.streams} streams with a workspace each (KeyEstimator(streams=...))" if args.streams > 1 else "1"},
        "roofline": {"bound": "mfma",
                     "kernel": ("conv_p2p_f16_ps_kernel (persistent 7x7 circular pitch convolution, 8 channels, f16 activations x f16 weights on "
                                "v_mfma_f32_16x16x32_f16 with f32 accumulation: 1 MFMA product per algorithmic MAC), 3 launches per step; the third "
                                "also runs the semitone conv and the octave maximum on its output tiles and writes only the folded maps")
                               if args.precision == "mixed" else
                               "conv_p2p_f16x3_kernel (the persistent 7x7 circular pitch convolution with f16 hi + lo operands: three v_mfma_f32_16x16x32_f16 products per algorithmic MAC, f32-equivalent to 2^-22; f32 NCHW in and out), 3 launches per step",
                     "achieved": round(achieved, 2) if achieved else None, "peak": DOM_PEAK, "unit": "TFLOP/s",
                     "frac": round(achieved / DOM_PEAK, 4) if achieved else None,
                     "mfma_products_per_mac": 1 if args.precision == "mixed" else 3,
                     "traffic": p2p_traffic,
                     "traffic_source": f"profiles/{traffic_src}: mean HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE (x2) / WRITE_SIZE" if p2p_traffic else traffic_stale,
                     # mean over the three launches: (1 CQT + 4 x 36-row up_sixth channels, f32 -> 8 channels, f16), (8 -> 8, f16 both), (8 f16 -> 8
                     # folded semitone channels of 12 rows, f32: the semitone conv and the octave max run inside the launch): the bytes this
                     # formulation has to move
                     "algorithmic_bytes_per_launch": B * T_FRAMES * ((4 * (P + 4 * 36) + 2 * 8 * P) + (2 * 8 * P + 2 * 8 * P) + (2 * 8 * P + 4 * 8 * 12)) // 3,
                     "avg_launch_ms": round(p2p_ms / p2p_n, 4) if p2p_n else None, "launches_per_step": launches_per_step,
                     "note": "measured in the timed region: with streams > 1 another step's kernels share the GPU with these launches" if args.streams > 1 else None,
                     "single_stream": {"achieved": round(achieved1, 2) if achieved1 else None,
                                       "frac": round(achieved1 / DOM_PEAK, 4) if achieved1 else None,
                                       "avg_launch_ms": round(p2p1_ms / p2p1_n, 4) if p2p1_n else None,
                                       "note": "same launches in the second, untimed pass: one step after the other on one stream"},
                     "algorithmic_flops_per_clip": 2 * P2P_MACS_PER_CLIP,
                     # `peak` is the data-sheet figure the contract asks for.  What a pure v_mfma_f32_16x16x32_f16 stream (4 waves per SIMD, nothing
                     # else issued) sustains on this part, measured by tools/micro/mfma_data_power.hip: 1 365 TFLOP/s with zero operands, 1 185 with
                     # random ones, whether the launch lasts 0.25 ms or 180 ms (profiles/r03_b_mfma_data_power.txt)
                     "measured_mfma_stream_tflops": {"zero_operands": 1365.0, "random_operands": 1185.0, "source": "profiles/r03_b_mfma_data_power.txt",
                                                     "frac_of_random": round(achieved / 1185.0, 4) if achieved else None}},
        "roofline_cqt": {"bound": "hbm", "kernels": " + ".join(sorted(k for k in prof_all if k.startswith("cqt_"))), "achieved": round(cqt_gbs, 1) if cqt_gbs else None,
                         "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(cqt_gbs / PEAK_HBM_GBS, 4) if cqt_gbs else None,
                         "traffic": cqt_traffic, "algorithmic_bytes_per_step": CQT_BYTES_PER_CLIP * B, "algorithmic_bytes_per_clip": CQT_BYTES_PER_CLIP,
                         "stage_ms_per_step": round(cqt_ms / args.steps, 4)},
        "kernel_ms_per_step": kernel_ms,
        "pipelined": {"streams": args.pipelined_streams, "value": round(clips / dt_pipe, 1), "unit": "clips/s",
                      "ms_per_step": round(dt_pipe / args.steps * 1e3, 4),
                      "note": "separately timed pass of the same K steps, issued round-robin on that many streams with a workspace each "
                              "(KeyEstimator(streams=...)): one batch's VALU / HBM-bound CQT runs under another's MFMA-bound convolutions; "
                              "not the headline value because overlapping kernels blur the per-kernel durations the roofline is made of"} if dt_pipe else None,
        "net_algorithmic_tflops": round(2.0 * NET_MACS_PER_CLIP * value / world / 1e12, 2),
        "sustained": sustained, "parity": parity, "max_rel_err": parity["max_rel_err"] if parity else None, "mirex": mirex,
    }
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(sd)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
