"""Host-side logic that needs no GPU: synthetic clips, label encoding, drop-in class surface, loud failure."""
from argparse import Namespace

import numpy as np
import pytest
import torch

import ake_amd
from ake_amd import synthetic
from ake_amd.KeyDataset import labels_for_signature, signature_id
from conftest import golden_state_dict


def test_synthetic_clip_is_seeded_and_labelled():
    y1, l1 = synthetic.make_clip(5, 22050)
    y2, _ = synthetic.make_clip(5, 22050)
    assert y1.dtype == np.float32 and np.array_equal(y1, y2) and abs(np.abs(y1).max() - 0.9) < 1e-6
    assert not np.array_equal(y1, synthetic.make_clip(6, 22050)[0])
    assert l1["key_signature_id"].argmax() == 5 and l1["tonic_labels"].argmax() == 5 and l1["genre"].argmax() == 5
    # F minor (id 5) shares its pitch classes with Ab major (id 20)
    assert np.array_equal(l1["key_labels"], synthetic.key_pitch_classes(20)) and l1["key_labels"].sum() == 7
    ys, labs = synthetic.make_batch(range(3), 4410)
    assert ys.shape == (3, 4410) and labs["key_signature_id"].shape == (3, 24)


def test_label_encoding_matches_the_reference_table(gold_mirex):
    table = gold_mirex["table"]
    # KeyDataset.py:443-447: key_labels = KEY_SIGNATURE_MAP[keys.index(name) % 21]; C major -> row 7, A minor -> row 7
    kl, ks, g, t = labels_for_signature(signature_id("C major"), 3, True)
    assert np.array_equal(kl.numpy(), table[7]) and ks.argmax() == 12 and t.argmax() == 0 and g.shape == (11,) and g[3] == 1
    kl, ks, g, t = labels_for_signature(signature_id("A minor"), None, False)
    assert np.array_equal(kl.numpy(), table[7]) and ks.argmax() == 9 and t.argmax() == 9 and g.shape == (8,) and g.sum() == 0
    assert signature_id("F# minor") == signature_id("Gb minor") == 6
    for sig in range(24):                                        # every key's set is a row of the table
        assert any(np.array_equal(labels_for_signature(sig, None, False)[0].numpy(), r) for r in table)


def test_drop_in_state_dict_and_init(gold_default):
    """Same keys/shapes as the reference state_dict, and the same torch seed gives the reference's initial weights."""
    torch.manual_seed(0)
    net = ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(genre=True))
    sd = net.state_dict()
    ref = golden_state_dict(gold_default)
    assert set(sd) == set(ref)
    for k in sd:
        assert sd[k].shape == ref[k].shape, k
    conv_keys = [k for k in sd if sd[k].dim() == 4 or (k.endswith(".bias") and (k[:-5] + ".weight") in sd and sd[k[:-5] + ".weight"].dim() == 4)]
    assert len(conv_keys) == 36
    for k in conv_keys:
        assert torch.equal(sd[k], ref[k]), k                    # fixture convs are the seed-0 reference init
    net.load_state_dict(ref, strict=True)                       # eval.py:115
    assert sum(p.numel() for p in net.parameters()) == 167031
    k2 = ake_amd.PitchClassNet(360, 12, 2, 7, Namespace(genre=False))
    assert not any(k.startswith("genre_classifier") for k in k2.state_dict())


def test_forward_fails_loudly_without_a_gpu_tensor():
    net = ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(genre=True)).eval()
    with pytest.raises(ake_amd._lib.AkeError, match="No CPU fallback"):
        net(torch.zeros(1, 1, 288, 76), None)
    with pytest.raises(ake_amd._lib.AkeError, match="No CPU fallback"):
        net.train()(torch.zeros(1, 1, 288, 76), None)            # the train-mode forward is HIP-only as well


@pytest.mark.parametrize("flag", ["only_semitones"])
def test_variant_flags_raise(flag):
    with pytest.raises(NotImplementedError):
        ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(**{flag: True}))


@pytest.mark.parametrize("flag", ["resblock", "pc2p_mem", "p2pc_conv", "stay_sixth"])
def test_variant_state_dict_equals_reference(flag, gold_resblock, gold_pc2pmem, gold_p2pcconv, gold_staysixth):
    """--resblock / --pc2p_mem: keys, order and shapes of the state_dict equal the reference's (strict load of a reference checkpoint)."""
    net = ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(genre=True, **{flag: True}))
    gold = {"resblock": gold_resblock, "pc2p_mem": gold_pc2pmem, "p2pc_conv": gold_p2pcconv, "stay_sixth": gold_staysixth}[flag]
    ref = {k[3:]: gold[k] for k in gold.files if k.startswith("sd/")}
    sd = net.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    assert all(tuple(sd[k].shape) == ref[k].shape for k in ref)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in ref.items()}, strict=True)


def test_denseblock_state_dict_equals_reference(gold_denseblock):
    """--denseblock: keys, order and shapes of the state_dict equal the reference's (its own n_filters / conv_layers from the fixture)."""
    import json
    opt = Namespace(**json.loads(str(gold_denseblock["opt"])))
    net = ake_amd.PitchClassNet(288, 12, opt.num_layers, opt.kernel_size, opt)
    ref = {k[3:]: gold_denseblock[k] for k in gold_denseblock.files if k.startswith("sd/")}
    sd = net.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    assert all(tuple(sd[k].shape) == ref[k].shape for k in ref)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in ref.items()}, strict=True)
    # default widths: the channel algebra of models.py:267-278, 678-689
    big = ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(genre=True, denseblock=True))
    assert big.state_dict()["model.1.pc2pc.layer.0.denselayer3.conv2.conv2d.weight"].shape == (4, 76, 12, 7)
    assert big.state_dict()["key_classifier.0.conv2d.weight"].shape == (102, 51, 12, 7)
    with pytest.raises(NotImplementedError):
        ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(denseblock=True, resblock=True))


def test_local_flag_keeps_the_state_dict_and_sets_the_window():
    """--local (models.py:720-722): parameter-free pooling after the heads => same state_dict keys; W = frames * loc_window_size - 12."""
    a = ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(genre=True))
    b = ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(genre=True, local=True, frames=5, loc_window_size=10, head_layers=2))
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())
    assert b.local and b.local_window == 38 and a.local_window == 0
    with pytest.raises(ValueError):
        ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(local=True, frames=1, loc_window_size=10))


def test_configure_optimizers_contract():
    net = ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(genre=True, lr=3e-4, reg=0, gamma=0.96))
    (opt,), (sched,) = net.configure_optimizers()                # models.py:1017-1027
    assert isinstance(opt, torch.optim.Adam) and opt.defaults["lr"] == 3e-4 and opt.defaults["betas"] == (0.9, 0.999)
    assert isinstance(sched, torch.optim.lr_scheduler.ExponentialLR) and sched.gamma == 0.96
    assert sum(p.numel() for g in opt.param_groups for p in g["params"]) == 167031
