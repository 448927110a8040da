"""The CPU oracle (oracle/pcnet_oracle.py) against the fixtures generated from the real reference."""
import json
from argparse import Namespace

import numpy as np
import pytest
import torch

from conftest import golden_state_dict
from oracle import mirex_oracle, pcnet_oracle


def test_default_outputs_fp64(gold_default):
    sd = golden_state_dict(gold_default, torch.float64)
    x = torch.from_numpy(gold_default["x"]).double()
    seq = torch.from_numpy(gold_default["seq_length"])
    k, t, g = pcnet_oracle.pcnet_forward(sd, x, seq)
    assert np.abs(k.numpy() - gold_default["key"]).max() <= 1e-12
    assert np.abs(t.numpy() - gold_default["tonic"]).max() <= 1e-12
    assert np.abs(g.numpy() - gold_default["genre"]).max() <= 1e-12
    k, t, g = pcnet_oracle.pcnet_forward(sd, x, None)
    assert np.abs(k.numpy() - gold_default["key_noseq"]).max() <= 1e-12
    assert np.abs(t.numpy() - gold_default["tonic_noseq"]).max() <= 1e-12
    assert np.abs(g.numpy() - gold_default["genre_noseq"]).max() <= 1e-12


def test_default_outputs_fp32_within_budget(gold_default):
    """fp32 arithmetic alone stays far inside the 1e-3 budget (sets expectations for the HIP path)."""
    sd = golden_state_dict(gold_default, torch.float32)
    k, t, g = pcnet_oracle.pcnet_forward(sd, torch.from_numpy(gold_default["x"]), torch.from_numpy(gold_default["seq_length"]))
    for got, name in ((k, "key"), (t, "tonic"), (g, "genre")):
        ref = gold_default[name]
        assert np.abs(got.numpy() - ref).max() / np.abs(ref).max() < 1e-4


def test_layer_taps(gold_default, gold_taps):
    sd = golden_state_dict(gold_default, torch.float64)
    taps = {}
    outs = pcnet_oracle.pcnet_forward(sd, torch.from_numpy(gold_taps["x"]).double(), torch.from_numpy(gold_taps["seq_length"]), taps=taps)
    names = [n[4:] for n in gold_taps.files if n.startswith("tap/")]
    assert len(names) == 16
    for n in names:
        ref = gold_taps["tap/" + n]            # stored as float32
        assert taps[n].shape == ref.shape
        assert np.abs(taps[n].numpy() - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), n
    for got, n in zip(outs, ("key", "tonic", "genre")):
        assert np.abs(got.numpy() - gold_taps[n]).max() <= 1e-12


def test_guard_octave_table_and_equivariance(gold_guard):
    """Rows of the (25,12) stack of equivariance_test.py:172-205, eval and train mode, and the roll identity."""
    sd = golden_state_dict(gold_guard, torch.float64)
    opt = Namespace(**json.loads(str(gold_guard["opt"])))
    assert opt.octaves == 10 and not opt.genre
    mel = gold_guard["mel"].astype(np.float64)
    mel_g = np.concatenate([np.zeros((36, 40)), mel, np.zeros((36, 40))])
    seq = torch.tensor(40).reshape(1, 1)

    def run(m, training):
        return pcnet_oracle.pcnet_forward(sd, torch.from_numpy(m).reshape(1, 1, 360, 40), seq, training=training)

    for training, kk, tt in ((False, "key_eval", "tonic_eval"), (True, "key_train", "tonic_train")):
        for s in (0, 1, 5, 12):
            k, t = run(mirex_oracle.mel_shifting_up(mel_g, s), training)
            assert np.abs(k[0].numpy() - gold_guard[kk][12 - s]).max() < 1e-11
            assert np.abs(t[0].numpy() - gold_guard[tt][12 - s]).max() < 1e-11
            k, t = run(mirex_oracle.mel_shifting_down(mel_g, s), training)
            assert np.abs(k[0].numpy() - gold_guard[kk][12 + s]).max() < 1e-11
    # transposition equivariance of the reference outputs themselves
    base_k, base_t = gold_guard["key_eval"][12], gold_guard["tonic_eval"][12]
    for s in range(1, 13):
        assert np.abs(gold_guard["key_eval"][12 - s] - np.roll(base_k, s)).max() < 1e-13
        assert np.abs(gold_guard["key_eval"][12 + s] - np.roll(base_k, -s)).max() < 1e-13
        assert np.abs(gold_guard["tonic_eval"][12 - s] - np.roll(base_t, s)).max() < 1e-13


def test_circular_roll_on_training_geometry(gold_default):
    """288-bin geometry: a circular roll by 3 bins (one semitone) rolls key/tonic exactly; genre is NOT invariant."""
    sd = golden_state_dict(gold_default, torch.float64)
    x = torch.from_numpy(gold_default["x"][:1]).double()
    k0, t0, g0 = pcnet_oracle.pcnet_forward(sd, x, None)
    k1, t1, g1 = pcnet_oracle.pcnet_forward(sd, torch.roll(x, 3, dims=2), None)
    assert (k1 - torch.roll(k0, 1, dims=1)).abs().max() < 1e-13
    assert (t1 - torch.roll(t0, 1, dims=1)).abs().max() < 1e-13
    assert (g1 - g0).abs().max() > 1e-9


def test_local_heads_against_reference_fixture(gold_default, gold_local):
    """--local (models.py:348, 394, 720-722, 805-810): no time pooling, MaxPool2d((1, 38), stride 1) after the key / tonic heads,
    per-frame outputs in the reference's reshape order; fixture from the reference run with opt.local on the default weights."""
    sd = golden_state_dict(gold_default, torch.float64)
    x = torch.from_numpy(gold_local["x"]).double()
    W = int(gold_local["window"])
    assert W == 5 * 10 - 2 * 6
    k, t, g = pcnet_oracle.pcnet_forward(sd, x, None, local_window=W)
    assert k.shape == (2, 120 - 12 - W + 1, 12) and g.shape == (2, 108, 11)
    assert np.abs(k.numpy() - gold_local["key"]).max() <= 1e-12
    assert np.abs(t.numpy() - gold_local["tonic"]).max() <= 1e-12
    assert np.abs(g.numpy() - gold_local["genre"]).max() <= 1e-12


def test_resblock_against_reference_fixture(gold_resblock):
    """--resblock (models.py:181-187, 218-224, 402-454): every stack = conv + BN + LeakyReLU, then conv_layers residual blocks
    x -> act(x + b2(conv2(act(b1(conv1(x)))))); fixture from the reference run with opt.resblock."""
    sd = golden_state_dict(gold_resblock, torch.float64)
    assert "model.1.p2p.layer.3.conv1.weight" in sd and "model.0.pc2pc.layer.5.conv2.conv2d.weight" in sd
    k, t, g = pcnet_oracle.pcnet_forward(sd, torch.from_numpy(gold_resblock["x"]).double(), torch.from_numpy(gold_resblock["seq_length"]))
    assert np.abs(k.numpy() - gold_resblock["key"]).max() <= 1e-12
    assert np.abs(t.numpy() - gold_resblock["tonic"]).max() <= 1e-12
    assert np.abs(g.numpy() - gold_resblock["genre"]).max() <= 1e-12


def test_pc2p_mem_against_reference_fixture(gold_pc2pmem):
    """--pc2p_mem (models.py:145-166, 333, 376-377): the up_sixth map is summed over its channels and added to the pitch stream (row r
    takes third-semitone index r // 8 -- the reference's reshape), the first pitch conv has prev_p input channels."""
    sd = golden_state_dict(gold_pc2pmem, torch.float64)
    assert sd["model.1.p2p.layer.0.weight"].shape == (8, 1, 7, 7)
    k, t, g = pcnet_oracle.pcnet_forward(sd, torch.from_numpy(gold_pc2pmem["x"]).double(), torch.from_numpy(gold_pc2pmem["seq_length"]))
    assert np.abs(k.numpy() - gold_pc2pmem["key"]).max() <= 1e-12
    assert np.abs(t.numpy() - gold_pc2pmem["tonic"]).max() <= 1e-12
    assert np.abs(g.numpy() - gold_pc2pmem["genre"]).max() <= 1e-12
    p = torch.arange(2 * 288 * 3, dtype=torch.float64).reshape(1, 2, 288, 3)
    ps = torch.ones(1, 4, 36, 3, dtype=torch.float64) * torch.arange(36, dtype=torch.float64).reshape(1, 1, 36, 1)
    out = pcnet_oracle.pitchclass2pitch_memory(p, ps)
    assert float((out - p)[0, 0, 17, 0]) == 2 * (17 // 8) and float((out - p)[0, 1, 287, 2]) == 2 * 35      # groups of 2 channels, row // 8


def test_p2pc_conv_against_reference_fixture(gold_p2pcconv):
    """--p2pc_conv (models.py:108-133, 316-317, 340-341): the octave fold as a dilated convolution + BN + LeakyReLU."""
    sd = golden_state_dict(gold_p2pcconv, torch.float64)
    assert sd["model.1.pool.conv.weight"].shape == (8, 8, 8, 1)
    k, t, g = pcnet_oracle.pcnet_forward(sd, torch.from_numpy(gold_p2pcconv["x"]).double(), torch.from_numpy(gold_p2pcconv["seq_length"]))
    assert np.abs(k.numpy() - gold_p2pcconv["key"]).max() <= 1e-12
    assert np.abs(t.numpy() - gold_p2pcconv["tonic"]).max() <= 1e-12
    assert np.abs(g.numpy() - gold_p2pcconv["genre"]).max() <= 1e-12


def test_stay_sixth_against_reference_fixture(gold_staysixth):
    """--stay_sixth (models.py:322-323, 336, 366-367, 371, 385): layer 0's semitone map becomes the pitch stream (96 rows); the later
    layers have no up_sixth / pool_semi and repeat the pitch classes directly."""
    sd = golden_state_dict(gold_staysixth, torch.float64)
    assert "model.1.up_sixth.weight" not in sd and "model.1.pool_semi.weight" not in sd and "model.0.pool_semi.weight" in sd
    k, t, g = pcnet_oracle.pcnet_forward(sd, torch.from_numpy(gold_staysixth["x"]).double(), torch.from_numpy(gold_staysixth["seq_length"]))
    assert np.abs(k.numpy() - gold_staysixth["key"]).max() <= 1e-12
    assert np.abs(t.numpy() - gold_staysixth["tonic"]).max() <= 1e-12
    assert np.abs(g.numpy() - gold_staysixth["genre"]).max() <= 1e-12


def test_denseblock_against_reference_fixture(gold_denseblock):
    """--denseblock (models.py:188-189, 225-226, 456-648; n_filters = 2, conv_layers = 2 to keep the fixture small -- make_golden.py also
    checks the default widths against the reference, without storing them): DenseNet-style stacks whose layers read the concatenation of
    everything before them through a BatchNorm of their own."""
    sd = golden_state_dict(gold_denseblock, torch.float64)
    assert sd["model.1.p2p.layer.0.denselayer2.conv1.weight"].shape == (6, 8, 1, 1)          # bn_size 3 x growth 2 <- 6 + 2 channels
    assert sd["model.1.pc2pc.layer.0.denselayer1.conv1.conv2d.weight"].shape == (14, 15, 12, 1)
    assert "model.1.p2p.layer.0.denselayer1.conv1.bias" not in sd                             # bias=False, models.py:464
    k, t, g = pcnet_oracle.pcnet_forward(sd, torch.from_numpy(gold_denseblock["x"]).double(), torch.from_numpy(gold_denseblock["seq_length"]))
    assert np.abs(k.numpy() - gold_denseblock["key"]).max() <= 1e-12
    assert np.abs(t.numpy() - gold_denseblock["tonic"]).max() <= 1e-12
    assert np.abs(g.numpy() - gold_denseblock["genre"]).max() <= 1e-12


def test_max_pool_quirk(gold_default):
    """--max_pool with seq_length: only sample 0 takes the max (models.py:764-785)."""
    sd = golden_state_dict(gold_default, torch.float64)
    x = torch.from_numpy(gold_default["x"]).double()
    seq = torch.from_numpy(gold_default["seq_length"])
    km, tm, _ = pcnet_oracle.pcnet_forward(sd, x, seq, max_pool=True)
    ka, ta, _ = pcnet_oracle.pcnet_forward(sd, x, seq)
    assert np.abs(tm[1:].numpy() - ta[1:].numpy()).max() == 0
    assert np.abs(tm[0].numpy() - ta[0].numpy()).max() > 1e-6


def test_running_stats_restatement_matches_torch_batchnorm():
    """oracle.update_running_stats == nn.BatchNorm2d's train-mode side effect (momentum 0.1, unbiased variance)."""
    torch.manual_seed(0)
    bn = torch.nn.BatchNorm2d(5).double().train()
    with torch.no_grad():
        bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0); bn.weight.normal_(); bn.bias.normal_()
    sd = {"p." + k: v.clone() for k, v in bn.state_dict().items()}
    x = torch.randn(3, 5, 4, 7, dtype=torch.float64) * 2 + 1
    y = bn(x)
    with pcnet_oracle.record_bn_stats() as rows:
        y2 = pcnet_oracle._bn(x, sd, "p.", training=True)
    pcnet_oracle.update_running_stats(sd, rows)
    assert torch.allclose(y, y2, atol=1e-12)
    for k, v in bn.state_dict().items():
        assert torch.allclose(sd["p." + k].double(), v.double(), atol=1e-12), k


@pytest.mark.parametrize("ksz", [3, 5])
def test_kernel_size_3_and_5_against_reference_fixture(ksz):
    """--kernel_size 3 / 5 (train_model.py:194; k x k pitch convs, 12 x k pitch-class convs and heads, (2, k) genre conv): the reference's own
    outputs for its own seeded weights, with and without seq_length (the heads shrink the frame count by k - 1 per layer, models.py:754-760)."""
    from conftest import load_golden
    gold = load_golden(f"pcnet_k{ksz}_T40.npz")
    sd = golden_state_dict(gold, torch.float64)
    assert sd["model.1.p2p.layer.0.weight"].shape[-2:] == (ksz, ksz) and sd["key_classifier.0.conv2d.weight"].shape[-2:] == (12, ksz)
    x = torch.from_numpy(gold["x"]).double()
    for seq, sfx in ((torch.from_numpy(gold["seq_length"]), ""), (None, "_noseq")):
        for got, name in zip(pcnet_oracle.pcnet_forward(sd, x, seq, kernel_size=ksz), ("key", "tonic", "genre")):
            assert np.abs(got.numpy() - gold[name + sfx]).max() <= 1e-12, (name, sfx)


def test_denseblock_training_step_against_reference_fixture():
    """--denseblock in train() mode: the loss and every parameter's gradient of the REFERENCE's own forward + autograd (float64, its
    checkpointed dense layers included) for its own seeded weights -- the restatement's autograd must give the same numbers."""
    import torch.nn.functional as F
    from conftest import load_golden
    gold = load_golden("pcnet_denseblock_train_T40.npz")
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in golden_state_dict(gold, torch.float64).items()}
    out = pcnet_oracle.pcnet_forward(sd, torch.from_numpy(gold["x"]).double(), torch.from_numpy(gold["seq_length"]), training=True)
    loss = (F.binary_cross_entropy(out[0], torch.from_numpy(gold["key_labels"])) + F.cross_entropy(out[1], torch.from_numpy(gold["tonic_idx"]))
            + 0.1 * F.cross_entropy(out[2], torch.from_numpy(gold["genre_idx"])))
    assert abs(float(loss.detach()) - float(gold["loss"])) < 1e-12
    loss.backward()
    names = [k[5:] for k in gold.files if k.startswith("grad/")]
    gmax = max(float(np.abs(gold["grad/" + k]).max()) for k in names)
    for k in names:
        ref = gold["grad/" + k]
        assert np.abs(sd[k].grad.numpy() - ref).max() <= 1e-6 * max(float(np.abs(ref).max()), 1e-9 * gmax), k
