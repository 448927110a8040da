"""PARITY (GPU): the HIP multirate CQT, through the C ABI, against the float64 direct-form oracle.

The CQT oracle is the build's own specification (parity with librosa is UNPINNED, oracle/cqt_oracle.py).
Tolerance: 1e-3 of the tensor's peak (north_star); the multirate evaluation's own error is ~1.5e-4
(tools/cqt_multirate_proto.py), so we assert 5e-4.
"""
import numpy as np
import pytest
import torch

import ake_amd
from ake_amd import synthetic
from conftest import rel_err
from oracle import cqt_oracle as O

pytestmark = pytest.mark.gpu
SR, HOP, DEV = 22050, 4410, "cuda:0"
TOL = 5e-4


@pytest.fixture(scope="module")
def plan():
    return ake_amd.CQTPlan(SR, HOP, 288, 36, device=DEV)


def test_short_clips_against_direct_form(plan):
    ys, _ = synthetic.make_batch(range(3), SR * 3)
    rng = np.random.default_rng(0)
    ys = np.concatenate([ys, rng.normal(0, 0.3, (1, SR * 3)).astype(np.float32)])        # + white noise (aliasing stress)
    got = plan.logmag(torch.from_numpy(ys).to(DEV)).cpu().numpy()
    assert got.shape == (4, 288, 16)
    for b in range(4):
        ref = O.cqt_logmag(ys[b], SR, HOP)
        assert rel_err(got[b], ref) < TOL, b


def test_full_length_clip_against_direct_form(plan):
    """One BASELINE-size clip: 15 s @ 22.05 kHz -> (288, 76)."""
    y, _ = synthetic.make_clip(7)
    got = plan.logmag(torch.from_numpy(y).to(DEV)).cpu().numpy()
    assert got.shape == (288, 76)
    ref = O.FastDirectCQT(SR, HOP, dtype=torch.float64)(y[None])[0].numpy()
    assert rel_err(got, ref) < TOL


@pytest.mark.parametrize("n", [4409, 4410, 4411, 30000, 70001])
def test_ragged_lengths_and_padding(plan, n):
    """Lengths around a hop boundary and clips shorter than the longest filter; frames past the clip are zero."""
    y, _ = synthetic.make_clip(2, n)
    T = O.n_frames(n, HOP)
    got = plan.logmag(torch.from_numpy(y).to(DEV), out_frames=T + 3).cpu().numpy()
    assert got.shape == (288, T + 3) and np.all(got[:, T:] == 0)
    assert rel_err(got[:, :T], O.cqt_logmag(y, SR, HOP)) < TOL


def test_other_rates_and_hops():
    for sr, frames, octaves in ((44100, 5, 8), (22050, 10, 8), (22050, 5, 7)):
        hop = ake_amd.hop_for(sr, frames)
        p = ake_amd.CQTPlan(sr, hop, 36 * octaves, 36, device=DEV)
        y, _ = synthetic.make_clip(4, sr * 2, sr)
        got = p.logmag(torch.from_numpy(y).to(DEV)).cpu().numpy()
        assert rel_err(got, O.cqt_logmag(y, sr, hop, n_bins=36 * octaves)) < TOL, (sr, frames, octaves)
    with pytest.raises(ake_amd._lib.AkeError, match="Nyquist"):
        ake_amd.CQTPlan(22050, 4410, 360, 36, device=DEV)                    # 10 octaves do not fit under 11 kHz


def test_full_size_batch_properties(plan):
    """B=64 full-length clips: scaling linearity of |C| and the hop-shift property (size-independent checks)."""
    audio, _ = synthetic.make_batch_device(range(64), DEV)
    out = plan.logmag(audio)
    assert out.shape == (64, 288, 76) and torch.isfinite(out).all()
    mag = torch.expm1(out)
    half = torch.expm1(plan.logmag(0.5 * audio))
    assert ((half - 0.5 * mag).abs().max() / mag.max()).item() < 1e-5
    # delaying a clip by exactly k hops moves its frames by k (interior frames, away from both edges).  Frame t and
    # frame t+k use different fractional-phase filter banks, so this holds to twice the multirate error, not to rounding.
    k = 3
    shifted = torch.zeros_like(audio)
    shifted[:, k * HOP:] = audio[:, :-k * HOP]
    out_s = plan.logmag(shifted)
    assert ((out_s[:, :, 12 + k:60] - out[:, :, 12:60 - k]).abs().max() / out.max()).item() < 2 * TOL
    # batch independence: a clip's rows do not depend on its neighbours
    solo = plan.logmag(audio[17:18])
    assert torch.equal(solo[0], out[17])
    # and 2 of the 64 against the oracle
    ref = O.FastDirectCQT(SR, HOP, dtype=torch.float64)(audio[[5, 40]].cpu().numpy()).numpy()
    assert rel_err(out[[5, 40]].cpu().numpy(), ref) < TOL


_CASCADE_SCRIPT = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import ake_amd
out = {}
g = torch.Generator().manual_seed(11)
for name, n, sr, hop, bins, bpo in (("full", 330750, 22050, 4410, 288, 36), ("ragged", 100003, 22050, 4410, 288, 36),
                                    ("short", 5000, 22050, 4410, 288, 36), ("nine_octaves", 132300, 44100, 4410, 324, 36),
                                    ("hop512", 40000, 22050, 512, 84, 12)):
    y = (torch.rand((3, n), generator=g) * 2 - 1).cuda()
    out[name] = ake_amd.cqt_logmag(y, sr, hop, n_bins=bins, bins_per_octave=bpo).cpu().numpy()
np.savez(sys.argv[2], **out)
"""


def test_fused_cascade_bit_identical_to_per_stage_kernels(tmp_path):
    """The one-pass streaming decimator (ring buffers, segment warm-up, sparse stores of the top levels) must hand the
    filter bank exactly the samples the plain one-kernel-per-stage cascade does: same arithmetic order -> equal bits.
    (AKE_CQT_LEGACY is read once per process, hence the two child processes; they run one after the other.)"""
    import os, subprocess, sys
    script = tmp_path / "cascade.py"
    script.write_text(_CASCADE_SCRIPT)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ("fused", "legacy"):
        env = dict(os.environ)
        env.pop("AKE_CQT_LEGACY", None)
        if mode == "legacy":
            env["AKE_CQT_LEGACY"] = "1"
        path = tmp_path / f"{mode}.npz"
        subprocess.run([sys.executable, str(script), repo, str(path)], check=True, env=env, timeout=600)
        res[mode] = np.load(path)
    for k in res["fused"].files:
        a, b = res["fused"][k], res["legacy"][k]
        assert a.shape == b.shape and np.isfinite(a).all()
        assert np.array_equal(a, b), (k, float(np.abs(a - b).max()))
