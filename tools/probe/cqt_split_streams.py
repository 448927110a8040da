#!/usr/bin/env python3
"""Does the CQT stage gain from running the cascade of one half-batch next to the filter bank of the other (two streams)?
The cascade is vector-issue bound, the bank memory-latency bound: if they overlap, a fork / join inside ake_cqt_logmag_f32 would pay.
    python3 tools/probe/cqt_split_streams.py [iters=50] [parts=2]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ake_amd
from ake_amd.cqt import CQTPlan

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B, n = 256, 330750
audio = torch.rand(B, n, device="cuda") - 0.5
full = CQTPlan(22050, 4410)
plans = [CQTPlan(22050, 4410) for _ in range(parts)]
streams = [torch.cuda.Stream() for _ in range(parts)]
T = full.num_frames(n)
out = torch.empty(B, 288, T, device="cuda")
step = B // parts


def timed(fn):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def one(): full.logmag(audio, out=out)


def seq():
    for i in range(parts): plans[i].logmag(audio[i * step:(i + 1) * step], out=out[i * step:(i + 1) * step])


def par():
    cur = torch.cuda.current_stream()
    for i, s in enumerate(streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            plans[i].logmag(audio[i * step:(i + 1) * step], out=out[i * step:(i + 1) * step])
    for s in streams: cur.wait_stream(s)


print(f"256 clips, one launch set: {timed(one):.4f} ms;  {parts} parts one stream: {timed(seq):.4f} ms;  {parts} parts on {parts} streams: {timed(par):.4f} ms")
