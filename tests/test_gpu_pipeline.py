"""GPU: the fused entry point, the KeyDataset drop-in and the validate loop."""
from argparse import Namespace

import numpy as np
import pytest
import torch

import ake_amd
from ake_amd import synthetic
from ake_amd.lightning_shim import Trainer
from conftest import REPO, golden_state_dict, rel_err
from oracle import cqt_oracle, mirex_oracle, pcnet_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def default_opt(**kw):
    o = dict(conv_layers=3, n_filters=4, head_layers=2, time_pool_size=2, genre=True, max_pool=False, frames=5, octaves=8,
             key_weight=1.0, tonic_weight=1.0, genre_weight=0.1, use_cos=False, no_ckpt=True, local=False,
             only_semitones=False, multi_scale=False)
    o.update(kw)
    return Namespace(**o)


@pytest.fixture(scope="module")
def net(gold_default):
    n = ake_amd.PitchClassNet(288, 12, 2, 7, default_opt())
    n.load_state_dict(golden_state_dict(gold_default), strict=True)
    return n.to(DEV).eval()


def test_pipeline_equals_cqt_then_forward_and_oracle(net, gold_default):
    est = ake_amd.KeyEstimator(net, 22050, 5)
    audio, _ = synthetic.make_batch(range(3), synthetic.N_SAMPLES)
    audio_d = torch.from_numpy(audio).to(DEV)
    key, tonic, genre = est(audio_d)
    mel = est.plan.logmag(audio_d)
    k2, t2, g2 = net(mel[:, None], torch.full((3,), 76, device=DEV))
    assert torch.equal(key, k2) and torch.equal(tonic, t2) and torch.equal(genre, g2)
    # end to end against the CPU oracle chain (direct-form CQT -> fp64 net)
    sd = golden_state_dict(gold_default, torch.float64)
    mel_ref = cqt_oracle.FastDirectCQT(22050, 4410, dtype=torch.float64)(audio)
    ref = pcnet_oracle.pcnet_forward(sd, mel_ref[:, None], torch.full((3,), 76))
    for a, b in zip((key, tonic, genre), ref):
        assert rel_err(a.cpu(), b) < 1e-3


def test_frames_major_path_is_bit_identical_and_skips_the_transpose(net):
    """for equal-length clips through the default net, ake_pipeline_forward_f32 leaves the CQT in the filter
    bank's [clip][frame][bin] order (ake_cqt_logmag_frames_major_f32) and the net's two readers transpose while staging
    (ake_pcnet_forward_frames_major_f32): no cqt_transpose_kernel, results equal to transpose + ake_pcnet_forward_f32 bit for bit."""
    import ctypes as C
    L = ake_amd._lib.lib()
    B, T = 64, 76
    audio = synthetic.make_batch_device(range(B), torch.device(DEV))[0]
    est = ake_amd.KeyEstimator(net, 22050, 5)
    mel = est.plan.logmag(audio)                                          # [B][288][T], through the transpose
    k2, t2, g2 = net(mel[:, None], torch.full((B,), T, device=DEV))
    assert L.ake_cqt_frames_major_supported(est.plan.handle) == 1
    assert L.ake_pcnet_accepts_frames_major(net._h, B, T) == 1
    assert L.ake_pcnet_accepts_frames_major(net._h, 2, T) == 0            # too few tiles for the persistent pitch conv: the plain route
    assert L.ake_pcnet_accepts_frames_major(net._h, B, T + 1) == 0        # odd frame count
    ake_amd._lib.prof_enable("", True)
    key, tonic, genre = est(audio)
    res = ake_amd._lib.prof_results()
    ake_amd._lib.prof_enable("", False)
    assert "cqt_transpose_kernel" not in res and "cqt_bank_bf16_kernel" in res
    assert torch.equal(key, k2) and torch.equal(tonic, t2) and torch.equal(genre, g2)
    # the two halves on their own
    mel_fm = torch.empty((B, T, 288), dtype=torch.float32, device=DEV)
    nbytes = L.ake_cqt_workspace_bytes(est.plan.handle, B, audio.shape[1])
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    stream = torch.cuda.current_stream().cuda_stream
    ake_amd._lib.check(L.ake_cqt_logmag_frames_major_f32(est.plan.handle, audio.data_ptr(), B, audio.shape[1], audio.stride(0), mel_fm.data_ptr(),
                                                         ws.data_ptr(), ws.numel(), stream), "ake_cqt_logmag_frames_major_f32")
    assert torch.equal(mel_fm.transpose(1, 2), mel)
    seq = torch.full((B,), T, dtype=torch.int64, device=DEV)
    outs = [torch.empty((B, n), dtype=torch.float32, device=DEV) for n in (12, 12, 11)]
    ws2 = torch.empty(L.ake_pcnet_workspace_bytes(net._h, B, T), dtype=torch.uint8, device=DEV)
    ake_amd._lib.check(L.ake_pcnet_forward_frames_major_f32(net._h, mel_fm.data_ptr(), B, T, seq.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(),
                                                            outs[2].data_ptr(), ws2.data_ptr(), ws2.numel(), stream), "ake_pcnet_forward_frames_major_f32")
    assert torch.equal(outs[0], k2) and torch.equal(outs[1], t2) and torch.equal(outs[2], g2)
    # a shape that cannot take the layout is refused instead of being read wrongly (2 clips: the per-tile pitch conv would run)
    assert L.ake_pcnet_forward_frames_major_f32(net._h, mel_fm.data_ptr(), 2, T, seq.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(),
                                                outs[2].data_ptr(), ws2.data_ptr(), ws2.numel(), stream) != 0
    assert b"frames-major" in L.ake_last_error()


def test_ragged_pipeline_equals_per_clip_pipeline(net):
    """ake_pipeline_forward_ragged_f32: every clip of a ragged batch is transformed on its own samples and pooled over its own frames
    (seq_length = 1 + n_i // hop, on the device): the same answer as the KeyDataset route -- per-clip CQT, zero padding to the longest
    clip, forward with seq_length (KeyDataset.py:242-256).  (Not the same as the clip in a batch of its own: the padded frames take
    part in the time-circular pitch convolutions, in the reference as here.)"""
    lens = [synthetic.N_SAMPLES, 22050 * 9 + 123, 22050 * 12, 22050 * 7]
    est = ake_amd.KeyEstimator(net, 22050, 5)
    rows = torch.zeros((len(lens), max(lens)))
    clips = []
    for i, n in enumerate(lens):
        y, _ = synthetic.make_clip(40 + i, n)
        clips.append(torch.from_numpy(y))
        rows[i, :n] = clips[-1]
        rows[i, n:] = 7.0                                                     # never read
    key, tonic, genre = est(rows.to(DEV), lengths=torch.tensor(lens))
    T = [1 + n // 4410 for n in lens]
    mel = torch.zeros((len(lens), 1, 288, max(T)), device=DEV)
    for i, y in enumerate(clips):
        mel[i, 0, :, :T[i]] = est.plan.logmag(y.to(DEV))
    k1, t1, g1 = net(mel, torch.tensor(T, device=DEV))
    assert (key - k1).abs().max() < 1e-5 and rel_err(tonic.cpu(), t1.cpu()) < 2e-5 and rel_err(genre.cpu(), g1.cpu()) < 2e-5
    assert (t1[1] - t1[0]).abs().max() > 0 and not torch.equal(mel[1], mel[0])  # the clips do differ


def test_two_streams_give_the_same_results(net):
    """KeyEstimator(streams=2): consecutive calls run on two side streams with a workspace each (one batch's CQT under another's
    convolutions); every call's outputs must equal the single-stream call's bit for bit, ragged batches included."""
    est1 = ake_amd.KeyEstimator(net, 22050, 5)
    est2 = ake_amd.KeyEstimator(net, 22050, 5, streams=2)
    batches = []
    for i in range(5):
        audio, _ = synthetic.make_batch_device(range(7 * i, 7 * i + 6), torch.device(DEV))
        batches.append(audio[:, : 330750 - 4410 * i].contiguous())
    lengths = torch.tensor([300000, 250000, 200000, 150000, 120000, 290000], device=DEV)
    outs = [est2(a) for a in batches] + [est2(batches[0], lengths)]
    est2.join()
    torch.cuda.synchronize()
    want = [est1(a) for a in batches] + [est1(batches[0], lengths)]
    for got, ref in zip(outs, want):
        for a, b in zip(got, ref):
            assert torch.equal(a, b)
    assert est2._slots[0]["stream"] is not None and est2._slots[1]["stream"] is not None
    assert est2._slots[0]["ws"].data_ptr() != est2._slots[1]["ws"].data_ptr()


def test_config5_sharded_equivariance_two_ranks():
    """BASELINE configs[4] / SURVEY 8d config 5: the 25 shifted guard-octave inputs dealt to two ranks (both on this GPU, gloo for
    the gather), roll identity <= 1e-5 and equality with the reference's table asserted by the tool itself."""
    import json, os, subprocess, sys
    env = dict(os.environ, AKE_REHEARSE_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(REPO, "tools", "config5_equivariance.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["pass"] and line["n_gpus"] == 2 and line["inputs"] == 25 and line["max_roll_identity_error"] <= 1e-5


def test_keydataset_item_contract_and_validate(net):
    opt = default_opt()
    ds = ake_amd.KeyDataset(True, opt)
    ds.import_data(ake_amd.SyntheticSineMixLoader(5, n_samples=22050 * 4),
                   ake_amd.SyntheticSineMixLoader(3, first=100, n_samples=22050 * 6, name="Synthetic long"), shuffle=False)
    assert len(ds) == 8 and ds.seq_length_max == 31
    item = ds[0]
    assert set(item) == {"mel", "key_labels", "tonic_labels", "key_signature_id", "genre", "seq_length"}   # KeyDataset.py:249-256
    assert item["mel"].shape == (1, 288, 31) and item["mel"].dtype == torch.float64 and item["seq_length"] == 21
    assert torch.all(item["mel"][:, :, 21:] == 0)
    assert item["key_labels"].shape == (12,) and item["tonic_labels"].shape == (12,) and item["key_signature_id"].shape == (24,)
    assert item["genre"].shape == (11,)
    ref = cqt_oracle.cqt_logmag(synthetic.make_clip(0, 22050 * 4)[0], 22050, 4410)
    assert rel_err(item["mel"][0, :, :21].numpy(), ref) < 5e-4
    # --genre off: zeros(8) label (KeyDataset.py:476)
    ds2 = ake_amd.KeyDataset(False, opt)
    ds2.import_data(ake_amd.SyntheticSineMixLoader(2, n_samples=22050 * 3), shuffle=False)
    assert ds2[1]["genre"].shape == (8,)
    # validate loop with the Lightning hook order; metrics equal the oracle's on the same outputs
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False)
    res = Trainer().validate(net, dataloaders=loader)[0]
    assert set(res) >= {"val_loss", "val_mirex_score", "val_accuracy", "val_accuracy_tonic", "val_accuracy_genre"}
    batch = next(iter(loader))
    out = net(batch["mel"].to(DEV), batch["seq_length"].to(DEV))
    m = mirex_oracle.mirex_score(batch["key_labels"].numpy(), out[0].cpu().numpy(), batch["tonic_labels"].numpy(),
                                 out[1].cpu().numpy(), batch["key_signature_id"].numpy())
    step = net.validation_step({k: v.to(DEV) if torch.is_tensor(v) else v for k, v in batch.items()}, 0)
    assert abs(float(step["val_mirex_score"]) - float(m[0])) < 1e-6
    from oracle import loss_oracle
    loss = loss_oracle.general_step_loss(out[0].cpu().numpy(), out[1].cpu().numpy(), out[2].cpu().numpy(), batch["key_labels"].numpy(),
                                         batch["tonic_labels"].numpy(), batch["genre"].numpy())
    assert abs(float(step["val_loss"]) - loss) < 1e-5


def test_kernel_timer_reports_the_launched_kernels(net):
    est = ake_amd.KeyEstimator(net, 22050, 5)
    audio = torch.from_numpy(synthetic.make_batch(range(2), 22050 * 6)[0]).to(DEV)
    est(audio)
    ake_amd._lib.prof_enable("", True)
    est(audio)
    res = ake_amd._lib.prof_results()
    ake_amd._lib.prof_enable("", False)
    assert {"cqt_bank_bf16_kernel", "cqt_cascade_kernel", "cqt_transpose_kernel", "conv_p2p_f16_kernel",
            "conv_pc_bf16_kernel/pc2pc", "conv_pc_bf16_kernel/head", "conv_head1_bf16_kernel"} <= set(res)
    assert "head_pool_kernel" not in res                                                   # the masked mean + sigmoid ride in conv_head1_bf16_kernel
    # Pitch2Pitch stack: all three convs on the f16 MFMA kernel (the 5-channel input is assembled channels-last, padded to 8)
    assert "conv_mfma_kernel/p2p" not in res and res["conv_p2p_f16_kernel"][1] == 3
    assert res["cqt_cascade_kernel"][1] == 1                                               # 7 decimation stages, one launch
    assert all(ms > 0 for ms, _ in res.values())


def test_pipeline_at_the_bench_batch(net, gold_default):
    """BASELINE configs[1] exactly as bench.py runs it: KeyEstimator on make_batch_device(range(256)).  The CQT bank tiles M over
    CLIPS (16 waves x 16 clips), so 256 clips is a shape of its own: the first and last clip of every 16-clip M-tile go through the
    CPU oracle chain (float64 direct-form CQT -> float64 network) at <= 1e-3, and the whole batch must be permutation-equivariant
    (clips are independent: reversing the batch reverses the rows, bit for bit)."""
    est = ake_amd.KeyEstimator(net, 22050, 5)
    audio, _ = synthetic.make_batch_device(range(256), torch.device(DEV))
    out = torch.cat(est(audio), 1)
    assert out.shape == (256, 35) and bool(torch.isfinite(out).all())
    pick = sorted({i for g in range(16) for i in (16 * g, 16 * g + 15)})
    sd = golden_state_dict(gold_default, torch.float64)
    with torch.no_grad():
        mel_ref = cqt_oracle.FastDirectCQT(22050, 4410, dtype=torch.float64)(audio[pick].cpu())
        ref = torch.cat(pcnet_oracle.pcnet_forward(sd, mel_ref[:, None], torch.full((len(pick),), 76)), 1)
    got = out[pick].cpu()
    for sl in (slice(0, 12), slice(12, 24), slice(24, 35)):
        assert rel_err(got[:, sl], ref[:, sl]) < 1e-3
    perm = torch.arange(255, -1, -1, device=DEV)
    out_p = torch.cat(est(audio[perm].contiguous()), 1)
    assert torch.equal(out_p, out[perm])
    # a different permutation that moves clips between M-tiles and between the network's persistent-kernel tiles
    perm2 = (torch.arange(256, device=DEV) * 37 + 11) % 256
    assert torch.equal(torch.cat(est(audio[perm2].contiguous()), 1), out[perm2])


def test_bench_starts_its_own_ranks():
    """`python3 bench.py --gpus 2` with NO launcher must start two rank processes itself and print an n_gpus: 2 line (both ranks on
    this GPU, gloo collectives: AKE_REHEARSE_ONE_GPU); a world size that differs from --gpus is a hard error."""
    import json, os, subprocess, sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["AKE_REHEARSE_ONE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "32", "--rotate", "2",
           "--sustained-seconds", "0.2", "--pipelined-streams", "0", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["clips_per_gpu"] == 32 and line["max_rel_err"] < 1e-3
    assert line["mirex"]["clips"] == 64 and line["sustained"]["steps"] >= 3
    # world size != --gpus: refused, no line
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--no-cpu-baseline"], env=env2,
                         capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in bad.stderr and not [l for l in bad.stdout.splitlines() if l.startswith("{")]
