"""PARITY (GPU): the HIP multirate CQT, through the C ABI, against the float64 direct-form oracle.

The CQT oracle is the build's own specification (parity with librosa is UNPINNED, oracle/cqt_oracle.py).
Tolerance: 1e-3 of the tensor's peak (north_star); the multirate evaluation's own error is ~1.5e-4
(tests/tools/cqt_multirate_proto.py), so we assert 5e-4.
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


def test_librosa09_filter_q_is_selectable():
    """requirements.txt:250 pins librosa 0.9.2, whose filter Q is 1 / (2^(1/36) - 1) instead of >= 0.10's; the choice is
    exposed through the Python layer (CQTPlan / get_plan / cqt_logmag / opt.cqt_q_mode / KeyEstimator) and checked against the
    direct-form oracle with the same constant.  The two Qs give measurably different features."""
    ys, _ = synthetic.make_batch(range(2), SR * 3)
    x = torch.from_numpy(ys).to(DEV)
    got09 = ake_amd.cqt.cqt_logmag(x, SR, HOP, 288, 36, q_mode=1).cpu().numpy()
    got10 = ake_amd.cqt.cqt_logmag(x, SR, HOP, 288, 36).cpu().numpy()
    for b in range(2):
        assert rel_err(got09[b], O.cqt_logmag(ys[b], SR, HOP, q_mode="librosa09")) < TOL
        assert rel_err(got10[b], O.cqt_logmag(ys[b], SR, HOP)) < TOL
    assert rel_err(got09, got10) > 5 * TOL
    from argparse import Namespace
    from ake_amd.KeyDataset import cqt_features
    mel = cqt_features(x, SR, Namespace(frames=5, octaves=8, cqt_q_mode=1))
    assert torch.equal(mel.cpu(), torch.from_numpy(got09))


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


def test_ragged_batch_equals_per_clip_transform(plan):
    """clips of different lengths in ONE call (ake_cqt_logmag_ragged_f32).  Whatever follows a clip in its row is
    never read (NaN there would poison the result); every clip equals its own single-clip transform and the direct-form oracle;
    frames past a clip's own length are zero (KeyDataset.py:245)."""
    lens = [70001, 70000, 4410 * 3 + 7, 50000, 4409, 30000]
    n_max = max(lens)
    rows = np.full((len(lens), n_max + 5), np.nan, np.float32)
    clips = []
    for i, n in enumerate(lens):
        y, _ = synthetic.make_clip(10 + i, n)
        clips.append(y)
        rows[i, :n] = y
    audio = torch.from_numpy(rows).to(DEV)[:, :n_max]                         # row stride n_max + 5
    T_max = O.n_frames(n_max, HOP)
    got = plan.logmag(audio, out_frames=T_max + 2, lengths=torch.tensor(lens)).cpu().numpy()
    assert got.shape == (len(lens), 288, T_max + 2) and np.isfinite(got).all()
    for i, n in enumerate(lens):
        T = O.n_frames(n, HOP)
        assert np.all(got[i, :, T:] == 0), i
        alone = plan.logmag(torch.from_numpy(clips[i]).to(DEV)).cpu().numpy()
        assert rel_err(got[i, :, :T], alone) < 1e-5, i
        assert rel_err(got[i, :, :T], O.cqt_logmag(clips[i], SR, HOP)) < TOL, i
    p1 = ake_amd.CQTPlan(SR, HOP, 288, 36, device=DEV, engine=1)             # the cross-check engines take equal-length batches only
    with pytest.raises(ake_amd._lib.AkeError, match="engine 3 or 5"):
        p1.logmag(audio, lengths=torch.tensor(lens))


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


ENGINE_CASES = (("full", 330750, 22050, 4410, 288, 36), ("ragged", 100003, 22050, 4410, 288, 36), ("short", 5000, 22050, 4410, 288, 36),
                ("nine_octaves", 132300, 44100, 4410, 324, 36), ("hop512", 40000, 22050, 512, 84, 12))


def _engine_outputs(engine, cases=ENGINE_CASES):
    from ake_amd.cqt import CQTPlan
    g = torch.Generator().manual_seed(11)
    out = {}
    for name, n, sr, hop, bins, bpo in cases:
        y = (torch.rand((3, n), generator=g) * 2 - 1).to(DEV)
        out[name] = CQTPlan(sr, hop, bins, bpo, engine=engine).logmag(y).cpu().numpy()
    return out


def test_fused_cascade_bit_identical_to_per_stage_kernels():
    """Engine 2 (one-pass streaming decimator: LDS level buffers, segment warm-up, sparse stores of the top levels) must
    hand the filter bank exactly the samples engine 1 (one kernel per stage) does: same arithmetic order -> equal bits."""
    a, b = _engine_outputs(2), _engine_outputs(1)
    for k in a:
        assert a[k].shape == b[k].shape and np.isfinite(a[k]).all()
        assert np.array_equal(a[k], b[k]), (k, float(np.abs(a[k] - b[k]).max()))


def test_bf16x3_bank_against_f32_bank():
    """Engine 3 (split-bf16 level signals + 3 bf16 MFMAs per product) against the exact-f32 engines: the split keeps 16
    mantissa bits per operand, so log-magnitudes agree to ~1e-5 of the largest value -- an order of magnitude inside the
    1.6e-4 the multirate design is specified to."""
    cases = [c for c in ENGINE_CASES if c[0] != "nine_octaves"]          # engine 3 covers <= 8 octaves
    a, b = _engine_outputs(3, cases), _engine_outputs(2, cases)
    for k in a:
        assert np.isfinite(a[k]).all()
        err = float(np.abs(a[k] - b[k]).max() / np.abs(b[k]).max())
        print(k, "bf16x3 vs f32 bank: rel err", err)
        assert err < 3e-5, (k, err)
    from ake_amd.cqt import CQTPlan
    with pytest.raises(ake_amd._lib.AkeError):
        CQTPlan(44100, 4410, 324, 36, engine=3)


ENGINE5_CASES = (("full", 3, 330750, 22050, 4410, 288, 36), ("b17", 17, 30000, 22050, 4410, 288, 36), ("tiny", 2, 5000, 22050, 4410, 288, 36),
                 ("hop2205", 3, 40000, 22050, 2205, 288, 36), ("seven_octaves", 2, 60000, 11025, 2205, 252, 36),
                 ("six_octaves", 2, 30000, 5512, 2300, 216, 36), ("b40_odd_stride", 40, 44113, 22050, 4411, 288, 36), ("hop512", 3, 40000, 22050, 512, 288, 36))


def test_streaming_mfma_cascade_engine5_against_the_exact_f32_engine():
    """Engine 5 (opt-in, round 3: the half-band stages of levels 0..4 as Toeplitz products on bf16 MFMA with the CLIPS as the N dimension --
    a wave streams 16 clips, the four stages chain through registers with two v_permlane swaps per tile pair, no LDS; deeper levels through
    engine 3's cascade on its f32 level 4; engine 3's filter bank) against engine 2, per octave: split operands keep 16 mantissa bits, so
    every octave agrees to 3e-5 of the tensor's peak (measured 5e-6 .. 9e-6).  Shapes: 6 to 8 octaves, batches that do not fill a 16-clip
    wave, a clip shorter than one segment (one wave per clip group), a row stride that is not a multiple of 4, small and large hops."""
    from ake_amd.cqt import CQTPlan
    g = torch.Generator().manual_seed(11)
    for name, B, n, sr, hop, bins, bpo in ENGINE5_CASES:
        y = (torch.rand((B, n), generator=g) * 2 - 1).to(DEV)
        ref = CQTPlan(sr, hop, bins, bpo, engine=2).logmag(y).cpu().numpy()
        got = CQTPlan(sr, hop, bins, bpo, engine=5).logmag(y).cpu().numpy()
        assert np.isfinite(got).all(), name
        peak = np.abs(ref).max()
        for o in range(bins // bpo):
            k0 = bins - bpo * (o + 1)
            err = float(np.abs(got[:, k0:k0 + bpo] - ref[:, k0:k0 + bpo]).max() / peak)
            assert err < 3e-5, (name, o, err)
    # ragged batch: equal to engine 3's
    lens = [70001, 70000, 4410 * 3 + 7, 50000, 4409, 30000]
    rows = torch.full((len(lens), max(lens) + 5), float("nan"))
    for i, n in enumerate(lens):
        rows[i, :n] = torch.rand(n, generator=g) * 2 - 1
    audio = rows.to(DEV)[:, :max(lens)]
    a = CQTPlan(SR, HOP, 288, 36, engine=3).logmag(audio, lengths=torch.tensor(lens)).cpu().numpy()
    b = CQTPlan(SR, HOP, 288, 36, engine=5).logmag(audio, lengths=torch.tensor(lens)).cpu().numpy()
    assert np.isfinite(b).all() and float(np.abs(a - b).max() / np.abs(a).max()) < 3e-5
    # fewer than six octaves: refused (engine 3 takes them); engine 4 of round 2 is gone
    with pytest.raises(ake_amd._lib.AkeError, match="engine 5"):
        CQTPlan(22050, 4410, 144, 36, engine=5)
    with pytest.raises(ake_amd._lib.AkeError, match="engine 4"):
        CQTPlan(22050, 4410, 288, 36, engine=4)
