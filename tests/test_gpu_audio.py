"""PARITY (GPU): the audio-preparation stage (channel selection / mono mix + polyphase resampling) through the
C ABI against its CPU restatement (pinned on scipy.signal.resample_poly, tests/test_oracle_misc.py), and the estimator's opt-in
wrap-at-the-clip's-true-end mode."""
from argparse import Namespace

import numpy as np
import pytest
import torch

import ake_amd
from ake_amd import synthetic
from conftest import golden_state_dict, rel_err
from oracle import cqt_oracle, pcnet_oracle, resample_oracle as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("rate_in,rate_out", [(44100, 22050), (48000, 22050), (24000, 22050), (11025, 22050), (16000, 22050)])
def test_resampler_against_the_oracle(rate_in, rate_out):
    rng = np.random.default_rng(rate_in)
    n = 30011
    x = rng.normal(0, 0.3, (3, 2, n)).astype(np.float32)                      # 3 stereo clips
    rs = ake_amd.Resampler(rate_in, rate_out, DEV)
    for channel in (0, 1, -1):
        got, lens = rs(torch.from_numpy(x).to(DEV), channel=channel)
        ref = np.stack([R.prepare(x[b], rate_in, rate_out, channel) for b in range(3)])
        assert got.shape == ref.shape and int(lens[0]) == ref.shape[1] == rs.out_len(n)
        assert rel_err(got.cpu().numpy(), ref) < 2e-6, (rate_in, channel)      # f32 accumulation of ~45 taps


def test_resampler_ragged_and_identity():
    rng = np.random.default_rng(3)
    lens = [20000, 12345, 7, 19999]
    rows = np.full((4, 1, 20000), np.nan, np.float32)
    for i, n in enumerate(lens):
        rows[i, 0, :n] = rng.normal(0, 0.3, n)
    rs = ake_amd.Resampler(48000, 22050, DEV)
    got, lo = rs(torch.from_numpy(rows).to(DEV), lengths=torch.tensor(lens))
    assert torch.isfinite(got).all()                                           # NaN behind a clip's end is never read
    for i, n in enumerate(lens):
        ref = R.resample_poly(rows[i, 0, :n], 48000, 22050)
        assert int(lo[i]) == len(ref)
        assert rel_err(got[i, :len(ref)].cpu().numpy(), ref) < 2e-6 and torch.all(got[i, len(ref):] == 0)
    # same rate, channel 0 = the reference's waveform[0] (KeyDataset.py:480): bit-identical
    x = torch.from_numpy(rng.normal(size=(2, 2, 1000)).astype(np.float32)).to(DEV)
    same, _ = ake_amd.Resampler(22050, 22050, DEV)(x, channel=0)
    assert torch.equal(same, x[:, 0])


@pytest.fixture(scope="module")
def net(gold_default):
    n = ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(genre=True))
    n.load_state_dict(golden_state_dict(gold_default), strict=True)
    return n.to(DEV).eval()


def test_estimator_takes_stereo_48k_audio(net, gold_default):
    """48 kHz stereo clips -> device mix-down + resampling -> CQT -> net, against the CPU chain (scipy-pinned resampler restatement ->
    float64 direct-form CQT -> float64 network)."""
    rng = np.random.default_rng(5)
    n48 = 48000 * 6
    t = np.arange(n48) / 48000.0
    x = np.stack([np.stack([0.3 * np.sin(2 * np.pi * (220.0 * (b + 1)) * t + ph) + 0.1 * np.sin(2 * np.pi * 1760.0 * t) + rng.normal(0, 0.003, n48)
                            for ph in (0.0, 1.0)]) for b in range(2)]).astype(np.float32)
    est = ake_amd.KeyEstimator(net, 22050, 5)
    got = est(torch.from_numpy(x).to(DEV), rate=48000, channel=-1)
    mono = np.stack([R.prepare(x[b], 48000, 22050, -1) for b in range(2)])
    mel = cqt_oracle.FastDirectCQT(22050, 4410, dtype=torch.float64)(mono)
    sd = golden_state_dict(gold_default, torch.float64)
    ref = pcnet_oracle.pcnet_forward(sd, mel[:, None], torch.full((2,), mel.shape[2]))
    for a, b in zip(got, ref):
        assert rel_err(a.cpu(), b) < 1e-3


def test_true_end_wrap_mode(net):
    """wrap_mode="true_end": every clip of a ragged batch gets the outputs it has ALONE (its time-circular convolutions wrap at its own
    last frame); the default mode keeps the reference's pad-to-the-longest behaviour, under which a short clip's outputs differ."""
    lens = [22050 * 9 + 123, synthetic.N_SAMPLES, 22050 * 9 + 4000, 22050 * 7, synthetic.N_SAMPLES - 5]
    rows = torch.zeros((len(lens), max(lens)))
    clips = []
    for i, n in enumerate(lens):
        y, _ = synthetic.make_clip(60 + i, n)
        clips.append(torch.from_numpy(y))
        rows[i, :n] = clips[-1]
    est_t = ake_amd.KeyEstimator(net, 22050, 5, wrap_mode="true_end")
    est_d = ake_amd.KeyEstimator(net, 22050, 5)
    got = est_t(rows.to(DEV), lengths=torch.tensor(lens))
    pad = est_d(rows.to(DEV), lengths=torch.tensor(lens))
    for i, y in enumerate(clips):
        alone = est_d(y[None].to(DEV))
        for a, b in zip(got, alone):
            assert rel_err(a[i:i + 1].cpu(), b.cpu()) < 2e-5, i
    assert (pad[1][3] - got[1][3]).abs().max() > 1e-4          # the 7 s clip: padded to 15 s it sees the wrap into zeros
    with pytest.raises(ValueError):
        ake_amd.KeyEstimator(net, 22050, 5, wrap_mode="nope")
