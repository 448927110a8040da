#!/usr/bin/env python3
"""Headline benchmark: clips/s (15 s @ 22.05 kHz) through HIP CQT + PitchClassNet forward.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

One "step" = one pass of the hot path (``ake_pipeline_forward_f32``: CQT -> seq_length fill ->
network) over B = 256 synthetic sine-mix clips that are already resident in HBM (BASELINE.json
configs[1]).  For N > 1 the driver launches this file under ``torch.distributed.run``; every rank
owns 256 clips of its own (weak scaling, clips are independent, no data-path collective) and the
step time is the max over ranks.  Rank 0 prints ONE JSON line.

Extra objects on the line (tier contract):
  roofline      dominant kernel (conv_p2p_bf16_kernel: the 7x7 circular pitch convolutions, 65 % of
                the MACs): algorithmic FLOPs (2 x MACs) of its launches / their hipEvent-measured
                duration, vs the 2.5 PFLOP/s dense bf16 MFMA peak of MI355X; the kernel multiplies
                split-bf16 operands (3 MFMA products per MAC), so `frac_of_split_ceiling` = 3 x frac
                is the fraction of what this formulation can reach.
  roofline_cqt  the CQT stage against the 8 TB/s HBM peak (1 410 552 algorithmic bytes per clip).
  cpu_baseline  the CPU oracle (direct-form CQT as BLAS matmuls + the float64 network, the
                reference's dtype) timed on this box's host cores on a bounded sample.
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
P2P_MACS_PER_CLIP = (5 * 8 + 8 * 8 + 8 * 8) * 49 * P * T_FRAMES        # 180 166 656 (three 7x7 convs: 5->8, 8->8, 8->8)
PEAK_BF16_TFLOPS = 2500.0
NET_MACS_PER_CLIP = 277_395_712
PEAK_FP32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 counter passes (tools/pmc_traffic.py; FETCH_SIZE x2 on gfx950, see
    DESIGN.md "Measurement").  bench.py cannot collect PMC counters itself (they need the profiler around the process), so the
    line carries the numbers of the newest profile in profiles/ that was taken with this same command and batch size."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, {}
    with open(files[-1]) as f:
        return os.path.basename(files[-1]), json.load(f)["kernels"]


def load_fixture_weights():
    gold = np.load(os.path.join(REPO, "tests", "golden", "pcnet_default.npz"))
    return {k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd/")}


def cpu_baseline(sd, n_clips=24, batch=8):
    """Oracle ("port") on the host cores: same synthetic clips, CQT then network (float64, the reference dtype)."""
    from ake_amd import synthetic
    from oracle import cqt_oracle, pcnet_oracle
    threads = torch.get_num_threads()
    audio, _ = synthetic.make_batch(range(n_clips))
    cqt = cqt_oracle.FastDirectCQT(SR, HOP, dtype=torch.float32)
    sd64 = pcnet_oracle.to_dtype(sd, torch.float64)
    seq = torch.full((batch,), T_FRAMES)
    with torch.no_grad():
        pcnet_oracle.pcnet_forward(sd64, cqt(audio[:1])[:, None].double(), seq[:1])       # warm-up
        t0 = time.perf_counter()
        t_cqt = 0.0
        for s in range(0, n_clips, batch):
            c0 = time.perf_counter()
            mel = cqt(audio[s:s + batch])
            t_cqt += time.perf_counter() - c0
            pcnet_oracle.pcnet_forward(sd64, mel[:, None].double(), seq[: mel.shape[0]])
        dt = time.perf_counter() - t0
    return {"value": round(n_clips / dt, 2), "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": f"{n_clips} synthetic 15 s clips, batches of {batch}: oracle direct-form CQT (fp32 BLAS matmuls, "
                      f"{t_cqt / dt:.0%} of the time) + oracle PitchClassNet forward in float64 (reference dtype), "
                      f"torch CPU ops on {threads} threads, {dt:.1f} s"}


def init_ranks(D):
    """(rank, world, local_rank).  AKE_REHEARSE_ONE_GPU=1 runs the N > 1 code path on a one-GPU box: every rank uses cuda:0
    and the collectives go through gloo (RCCL needs one GPU per rank) -- a functional rehearsal, never a measurement."""
    if os.environ.get("AKE_REHEARSE_ONE_GPU"):
        rank, world, _ = D.init_from_env("gloo")
        return rank, world, 0
    return D.init_from_env()


def train_main(args):
    """BASELINE configs[3]: B clips per GPU (log-CQT features resident, as the reference caches them, KeyDataset.py:154-192),
    one training step = train-mode forward + general_step loss + HIP backward + ONE all-reduce of the flat 668 KB gradient
    buffer + fused Adam.  Secondary line; the headline metric is the inference line printed without --train."""
    import ake_amd
    from ake_amd import distributed as D, synthetic
    rank, world, local_rank = init_ranks(D)
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

    def step(i):
        optim.zero_grad()
        net.training_step(batch, i)["loss"].backward()
        optim.grad_scale = D.all_reduce_gradients(net)
        optim.step()

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
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[3]: {B} clips per GPU (288x76 log-CQT resident), default PitchClassNet, local BatchNorm, "
                               f"one all-reduce of the flat gradient buffer per step, fused Adam lr 3e-4",
                   "clips_per_gpu": B, "parallelism": f"data-parallel x{world}"},
        "train_fp32_frac_of_peak": round(3 * 2.0 * NET_MACS_PER_CLIP * value / world / (PEAK_FP32_TFLOPS * 1e12), 4),
        "kernel_ms_per_step": kernel_ms}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU per step")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent steps in flight per GPU in the TIMED region: steps are issued round-robin on this many streams (own "
                         "workspace each), so one batch's CQT overlaps another's convolutions; 1 (default) = strictly one step after the "
                         "other, which keeps the per-kernel durations of the roofline clean")
    ap.add_argument("--pipelined-streams", type=int, default=2,
                    help="an extra, separately timed pass of the same steps with this many in flight, reported as 'pipelined' (0: skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train", action="store_true",
                    help="BASELINE configs[3] instead of the headline: one fwd + bwd + gradient all-reduce + fused Adam step per GPU batch")
    args = ap.parse_args()
    if args.train:
        return train_main(args)

    import ake_amd
    from ake_amd import distributed as D, synthetic

    rank, world, local_rank = init_ranks(D)
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the product path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    sd = load_fixture_weights()
    net = ake_amd.PitchClassNet(P, 12, 2, 7, Namespace(genre=True))
    net.load_state_dict(sd, strict=True)
    net = net.to(dev).eval()
    est = ake_amd.KeyEstimator(net, SR, FRAMES, streams=args.streams)
    est1 = est if args.streams == 1 else ake_amd.KeyEstimator(net, SR, FRAMES)      # second, untimed pass: one step after the other
    B = args.batch
    lo = rank * B                                                       # each rank synthesises its own clips
    audio, _ = synthetic.make_batch_device(range(lo, lo + B), dev)
    assert audio.shape == (B, N_SAMPLES)

    for _ in range(max(args.warmup, 1)):
        out = est(audio)
        if est1 is not est:
            est1(audio)
    est.join()
    torch.cuda.synchronize()
    # timed region: hipEvents (on the launch stream) bracket only the dominant kernel -- 3 launches per step
    ake_amd._lib.lib().ake_prof_reset()
    ake_amd._lib.prof_enable("conv_p2p_bf16_kernel", True)
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = est(audio)
    est.join()
    torch.cuda.synchronize()
    D.barrier()
    dt = time.perf_counter() - t0
    prof = ake_amd._lib.prof_results()
    # second, untimed pass, one step after the other on one stream with every kernel bracketed: per-kernel breakdown, the CQT stage's
    # roofline, and the dominant kernel's duration when nothing else shares the GPU with it
    ake_amd._lib.prof_enable("", True)
    for _ in range(args.steps):
        est1(audio)
    torch.cuda.synchronize()
    prof_all = ake_amd._lib.prof_results()
    ake_amd._lib.prof_enable("", False)
    dt = D.max_over_ranks(dt, dev)
    # third pass (timers off): the same steps with several in flight -- the throughput a serving loop over independent batches gets
    dt_pipe = None
    if args.pipelined_streams > 1 and args.streams == 1:
        estp = ake_amd.KeyEstimator(net, SR, FRAMES, streams=args.pipelined_streams)
        for _ in range(2 * args.pipelined_streams):
            estp(audio)
        estp.join()
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            estp(audio)
        estp.join()
        torch.cuda.synchronize()
        D.barrier()
        dt_pipe = D.max_over_ranks(time.perf_counter() - t0, dev)

    # result collection (outside the timed region): 35 floats per clip, rank order
    rows = D.gather_rows(torch.cat(out, 1), B * world)
    if rank != 0:
        return
    assert rows.shape == (B * world, 35) and bool(torch.isfinite(rows).all())

    clips = B * world * args.steps
    value = clips / dt
    # dominant kernel: the three 7x7 pitch convolutions (65 % of the network's MACs), conv_p2p_bf16_kernel
    p2p_ms, p2p_n = prof.get("conv_p2p_bf16_kernel", (0.0, 0))
    launches_per_step = p2p_n / args.steps if args.steps else 0
    p2p_flops = 2.0 * P2P_MACS_PER_CLIP * B * args.steps              # algorithmic: (5*8 + 8*8 + 8*8) * 49 MACs per position
    achieved = p2p_flops / (p2p_ms * 1e-3) / 1e12 if p2p_ms > 0 else None
    cqt_ms = sum(prof_all.get(k, (0.0, 0))[0] for k in prof_all if k.startswith("cqt_"))
    cqt_gbs = CQT_BYTES_PER_CLIP * B * args.steps / (cqt_ms * 1e-3) / 1e9 if cqt_ms > 0 else None
    kernel_ms = {k: round(v[0] / args.steps, 4) for k, v in sorted(prof_all.items(), key=lambda kv: -kv[1][0])}
    p2p1_ms, p2p1_n = prof_all.get("conv_p2p_bf16_kernel", (0.0, 0))
    achieved1 = p2p_flops / (p2p1_ms * 1e-3) / 1e12 if p2p1_ms > 0 else None
    traffic_src, traffic = pmc_traffic()
    p2p_traffic = cqt_traffic = None
    if B == 256 and traffic:
        bf_rows = [v for k, v in traffic.items() if k.startswith("conv_p2p_bf16")]
        p2p_traffic = round(sum(v["hbm_bytes"] * v["dispatches"] for v in bf_rows) / max(1, sum(v["dispatches"] for v in bf_rows))) if bf_rows else None
        cqt_traffic = sum(v["hbm_bytes"] for k, v in traffic.items() if k.startswith("cqt_")) or None
    line = {
        "metric": "clips/s (15 s @ 22.05 kHz), HIP CQT + PitchClassNet forward",
        "value": round(value, 1), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 storage and accumulation; the pitch / pitch-class convolutions and the CQT filter bank multiply as 3-term split-bf16 on MFMA (hi*hi + lo*hi + hi*lo, ~1e-5 relative to f32)", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: batch={B} synthetic 15 s sine-mix clips per GPU, HIP CQT (288 bins, hop 4410) "
                               f"+ default PitchClassNet inference (genre head on), audio resident in HBM",
                   "clips_per_gpu": B, "n_samples": N_SAMPLES, "frames": T_FRAMES, "weights": "tests/golden/pcnet_default.npz (seeded)",
                   "parallelism": f"clip-sharded x{world}, no data-path collective",
                   "streams": args.streams,
                   "steps_in_flight": f"{args.streams}: every step is the whole path over one batch; consecutive steps go round-robin to "
                                      f"{args.streams} streams with a workspace each (KeyEstimator(streams=...))" if args.streams > 1 else "1"},
        "roofline": {"bound": "mfma",
                     "kernel": "conv_p2p_bf16_ps_kernel (persistent 7x7 circular pitch convolution, 8 channels, split-bf16 operands on "
                               "v_mfma_f32_16x16x32_bf16 with f32 accumulation: 3 MFMA products per algorithmic MAC), 3 launches per step; the third "
                               "also runs the semitone conv on its output tile and writes only the semitone maps",
                     "achieved": round(achieved, 2) if achieved else None, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_BF16_TFLOPS, 4) if achieved else None,
                     "mfma_products_per_mac": 3,
                     "frac_of_split_ceiling": round(3 * achieved / PEAK_BF16_TFLOPS, 4) if achieved else None,
                     "traffic": p2p_traffic,
                     "traffic_source": f"profiles/{traffic_src}: mean HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE (x2) / WRITE_SIZE" if p2p_traffic else None,
                     # mean over the three launches: (1 CQT + 4 x 36-row up_sixth channels -> 8 channels), (8 -> 8), (8 -> 8 semitone channels of P / 3 rows)
                     "algorithmic_bytes_per_launch": B * T_FRAMES * 4 * ((P + 4 * 36 + 8 * P) + (8 * P + 8 * P) + (8 * P + 8 * P // 3)) // 3,
                     "avg_launch_ms": round(p2p_ms / p2p_n, 4) if p2p_n else None, "launches_per_step": launches_per_step,
                     "note": "measured in the timed region: with streams > 1 another step's kernels share the GPU with these launches" if args.streams > 1 else None,
                     "single_stream": {"achieved": round(achieved1, 2) if achieved1 else None,
                                       "frac": round(achieved1 / PEAK_BF16_TFLOPS, 4) if achieved1 else None,
                                       "avg_launch_ms": round(p2p1_ms / p2p1_n, 4) if p2p1_n else None,
                                       "note": "same launches in the second, untimed pass: one step after the other on one stream"},
                     "algorithmic_flops_per_clip": 2 * P2P_MACS_PER_CLIP},
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
    }
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(sd)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
