#!/usr/bin/env python3
"""Design study: a TWO-PATH decimator cascade for the device CQT.  Stage s produces the signal the octave's
filter bank reads with the 47-tap half-band (as today), but the signal that only FEEDS the next stage with a shorter half-band: what
aliases into the bands the deeper octaves keep lies in [0.84, 1.0] of the stage's Nyquist, so the transition band may be 0.68 wide.
Numpy model against the direct-form oracle; prints the same error measures as cqt_multirate_proto.py.
Result (3 s clips): a 19-tap (Kaiser beta 9) or 23-tap (beta 10) pass-through filter in the first two or three stages leaves the error
where it is (white noise 7.8e-5 .. 1.0e-4, chirp 1.55e-4 of the log-magnitude range; 15 taps do not: 4e-4 .. 8e-4).  NOT built: only
stages 0 and 1 have pass-through samples that no window reads (14 % / 27 % of their outputs are window samples, 54 % at stage 2), a
wave of the cascade kernel covers 512 / 1024 full-rate samples of a 4410-sample hop, so per-wave divergence leaves stage 0 with -21 %
of its FIR instructions and stage 1 with none; a compacted extra phase for the window samples would save ~19 % of the FIR instructions
(~10 % of the kernel, which also moves 521 MB at 3.8 TB/s) for one more barrier per tick.

    python3 tests/tools/cqt_two_path_proto.py"""
import math, sys, os, importlib
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests", "tools"))
from oracle import cqt_oracle as O
import cqt_multirate_proto as M
synthetic = importlib.import_module("audio-key-estimation_amd.synthetic")

def multirate2(y, sr, hop, hq, hc, n_cheap, n_bins=288, bpo=36, dtype=np.float64):
    """two-path cascade: stage s (level s -> s+1) produces the QUALITY signal of level s+1 with hq from the pass-through signal of
    level s, and the PASS-THROUGH signal of level s+1 with hc (s < n_cheap) or hq."""
    y = np.asarray(y, dtype); n = len(y); T = O.n_frames(n, hop)
    freqs = O.cqt_frequencies(n_bins, bpo); lengths = O.cqt_lengths(sr, n_bins, bpo)
    n_oct = math.ceil(n_bins / bpo)
    out = np.zeros((n_bins, T), dtype=np.complex128)
    yp, yp_lo = y.astype(np.float64), 0          # pass-through signal of the current level
    yq, yq_lo = yp, yp_lo                         # quality signal of the current level
    chain = []                                    # filters applied to reach the quality signal of the current level
    for o in range(n_oct):
        dec = 2 ** o
        ks = range(max(0, n_bins - bpo * (o + 1)), n_bins - bpo * o)
        lo_min = min(math.floor(-lengths[k] / 2.0) for k in ks)
        Uh = math.ceil(-lo_min / dec) + 1
        u = np.arange(-Uh, Uh + 1, dtype=np.float64)
        ypad = np.concatenate([np.zeros(Uh + 2), yq, np.zeros(Uh + 2)]).astype(dtype)
        off = Uh + 2 - yq_lo
        for t in range(T):
            c = t * hop; c_int, ph = divmod(c, dec)
            pos = dec * u - ph
            seg = ypad[c_int + off - Uh: c_int + off + Uh + 1]
            if len(seg) < len(u): seg = np.concatenate([seg, np.zeros(len(u) - len(seg), dtype)])
            for k in ks:
                lo = math.floor(-lengths[k] / 2.0); L = math.floor(lengths[k] / 2.0) - lo
                inside = (pos >= lo) & (pos <= lo + L)
                w = np.where(inside, 0.5 - 0.5 * np.cos(2 * np.pi * (pos - lo) / L), 0.0)
                g = (dec * math.sqrt(lengths[k]) / (L / 2.0)) * w * np.exp(-2j * np.pi * freqs[k] * pos / sr)
                gain = 1.0
                for s, h in enumerate(chain):
                    wn = 2 * np.pi * freqs[k] / (sr / 2 ** s); hl = (len(h) - 1) // 2
                    gain *= abs(np.sum(h * np.exp(-1j * wn * np.arange(-hl, hl + 1))))
                g = g / gain
                out[k, t] = np.dot(seg, g.real.astype(dtype)) + 1j * np.dot(seg, g.imag.astype(dtype))
        if o + 1 < n_oct:
            yq, yq_lo = M.decimate(yp, yp_lo, hq)
            hp = hc if o < n_cheap else hq
            pass_chain = chain[:-1] if chain else []
            # filters on the way to the NEXT level's quality signal: the pass-through filters of all earlier stages, then hq
            new_pass = ([] if o == 0 else prev_pass) + [hp]
            chain = ([] if o == 0 else prev_pass) + [hq]
            prev_pass = new_pass
            yp, yp_lo = M.decimate(yp, yp_lo, hp) if hp is not hq else (yq, yq_lo)
    return out

sr, hop, n = 22050, 4410, 22050 * 3
rng = np.random.default_rng(0)
clips = {"sine-mix": synthetic.make_clip(3, n)[0].astype(np.float64), "white": rng.normal(0, 0.3, n),
         "chirp": 0.8 * np.sin(2 * np.pi * (30 * np.arange(n) / sr + 0.5 * (10000 / 3) * (np.arange(n) / sr) ** 2))}
hq = M.kaiser_halfband(23, 8.0)
for name, y in clips.items():
    ref = O.cqt_complex(y, sr, hop); lref = np.log1p(np.abs(ref))
    for (hl, beta, nc) in [(23, 8.0, 0), (7, 7.0, 3), (9, 9.0, 2), (9, 9.0, 3), (11, 10.0, 3), (15, 7.0, 3)]:
        hc = M.kaiser_halfband(hl, beta)
        got = multirate2(y, sr, hop, hq, hc, nc)
        lg = np.log1p(np.abs(got))
        print(f"{name:9s} cheap Hh={hl:2d} beta={beta:3.1f} stages<{nc}: |C| err {np.max(np.abs(np.abs(got)-np.abs(ref)))/np.max(np.abs(ref)):.2e}  log1p rel {np.max(np.abs(lg-lref))/np.max(lref):.2e}", flush=True)
