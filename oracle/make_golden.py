#!/usr/bin/env python3
"""Generate the golden fixtures under ``tests/golden/`` from the REAL reference.

Run in the build container only (needs /root/reference):

    python3 -B oracle/make_golden.py

It imports the reference's ``models.py`` unmodified (``oracle/_refstubs.py``), runs
it on CPU in float64 on seeded inputs, checks the restatements in ``oracle/``
against it, and writes inputs + expected outputs (numbers only -- no reference
source text) as small ``.npz`` files.  Weights and inputs are rounded to float32
first, so the float32 HIP path and the float64 reference consume bit-identical
values.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
from argparse import Namespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, "tests", "golden")

from oracle import _refstubs  # noqa: E402

models = _refstubs.install()
import torch  # noqa: E402

from oracle import loss_oracle, mirex_oracle, pcnet_oracle  # noqa: E402

torch.set_grad_enabled(False)


def default_opt(**kw):
    """Flag defaults of train_model.py:166-237."""
    o = dict(conv_layers=3, n_filters=4, num_layers=2, kernel_size=7, resblock=False, denseblock=False,
             stay_sixth=False, only_semitones=False, p2pc_conv=False, pc2p_mem=False, local=False,
             time_pool_size=2, head_layers=2, genre=True, frames=5, loc_window_size=10, max_pool=False,
             octaves=8, key_weight=1.0, tonic_weight=1.0, genre_weight=0.1, use_cos=False, acc_grad=8,
             window_size=592, no_ckpt=True, lr=3e-4, reg=0, gamma=0.96)
    o.update(kw)
    return Namespace(**o)


def build_reference_net(opt, seed):
    """Reference net with seeded weights; BN affine + running stats randomised so that
    BN folding is exercised (a fresh BN is the identity in eval mode)."""
    torch.manual_seed(seed)
    net = models.PitchClassNet(opt.octaves * 36, 12, opt.num_layers, opt.kernel_size, opt,
                               window_size=opt.window_size).double()
    g = torch.Generator().manual_seed(seed + 1)
    sd = net.state_dict()
    for name, mod in net.named_modules():      # named_modules(): .modules is shadowed (models.py:673)
        if isinstance(mod, torch.nn.BatchNorm2d):
            c = sd[name + ".weight"].shape
            sd[name + ".running_mean"] = torch.randn(c, generator=g, dtype=torch.float64) * 0.2
            sd[name + ".running_var"] = torch.rand(c, generator=g, dtype=torch.float64) * 1.5 + 0.5
            sd[name + ".weight"] = torch.rand(c, generator=g, dtype=torch.float64) + 0.5
            sd[name + ".bias"] = torch.randn(c, generator=g, dtype=torch.float64) * 0.1
    # round to float32 so both sides read identical numbers
    sd = {k: (v.float().double() if v.is_floating_point() else v) for k, v in sd.items()}
    net.load_state_dict(sd, strict=True)
    return net, sd


def sd_to_npz(sd):
    return {"sd/" + k: (v.numpy().astype(np.float32) if v.is_floating_point() else v.numpy()) for k, v in sd.items()}


def sha256(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def check(name, a, b, tol):
    err = float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) if np.size(a) else 0.0
    status = "ok" if err <= tol else "MISMATCH"
    print(f"  [{status}] {name}: max|oracle-reference| = {err:.3e} (tol {tol:g})")
    if err > tol:
        raise SystemExit(f"oracle restatement disagrees with the reference on {name}")
    return err


def main():
    os.makedirs(GOLD, exist_ok=True)
    report = {"reference_sha256": {f: sha256(os.path.join(_refstubs.REFERENCE_ROOT, f)) for f in
                                   ("models.py", "KeyDataset.py", "equivariance_test.py", "utils/key_signatures.py",
                                    "train_model.py", "eval.py")},
              "torch": torch.__version__, "checks": {}}

    # ---------------------------------------------------------------- A: default config, B=4, T=76
    print("A: default PitchClassNet (288 bins, genre head), B=4, T=76")
    opt = default_opt()
    net, sd = build_reference_net(opt, seed=0)
    net.eval()
    g = torch.Generator().manual_seed(100)
    x = (torch.rand((4, 1, 288, 76), generator=g) * 2.5).float()       # log1p|CQT| range
    x[2, :, :, 60:] = 0                                                # zero-padded shorter clips,
    x[3, :, :, 41:] = 0                                                # as KeyDataset.py:245 builds them
    seq = torch.tensor([76, 76, 60, 41])
    key, tonic, genre = net(x.double(), seq)
    key_n, tonic_n, genre_n = net(x.double(), None)
    ok, ot, og = pcnet_oracle.pcnet_forward(sd, x.double(), seq)
    report["checks"]["A_key"] = check("key", ok, key, 1e-12)
    report["checks"]["A_tonic"] = check("tonic", ot, tonic, 1e-12)
    report["checks"]["A_genre"] = check("genre", og, genre, 1e-12)
    ok2, ot2, og2 = pcnet_oracle.pcnet_forward(sd, x.double(), None)
    check("key (no seq_length)", ok2, key_n, 1e-12)
    check("tonic (no seq_length)", ot2, tonic_n, 1e-12)
    check("genre (no seq_length)", og2, genre_n, 1e-12)
    np.savez_compressed(os.path.join(GOLD, "pcnet_default.npz"),
                        opt=json.dumps(vars(opt)), x=x.numpy(), seq_length=seq.numpy(),
                        key=key.numpy(), tonic=tonic.numpy(), genre=genre.numpy(),
                        key_noseq=key_n.numpy(), tonic_noseq=tonic_n.numpy(), genre_noseq=genre_n.numpy(),
                        **sd_to_npz(sd))

    # ---------------------------------------------------------------- B: per-layer taps, B=1, T=28
    print("B: per-layer activations, B=2, T=28 (same weights)")
    xb = (torch.rand((2, 1, 288, 28), generator=g) * 2.5).float()
    taps_ref = {}
    hooks = []
    wanted = {}
    for name, mod in net.named_modules():           # NOT net.modules(): shadowed by a list (models.py:673)
        leaf = name.split(".")[-1]
        if isinstance(mod, torch.nn.LeakyReLU) or leaf in ("pool", "time_pool_pc") or \
                (name.endswith("_classifier.3") or name.endswith("_classifier.3.conv2d")):
            wanted[name] = mod
    for name, mod in wanted.items():
        hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: taps_ref.__setitem__(name, o.detach().clone())))
    outs_b = net(xb.double(), torch.tensor([28, 28]))
    for h in hooks:
        h.remove()
    taps_orc = {}
    outs_o = pcnet_oracle.pcnet_forward(sd, xb.double(), torch.tensor([28, 28]), taps=taps_orc)
    name_map = {"model.0.pool": "model.0.pool", "model.1.pool": "model.1.pool",
                "model.1.time_pool_pc": "model.1.time_pool_pc", "model.1.up_sixth_a": "model.1.up_sixth_a"}
    for i in (2, 5, 8):
        name_map[f"model.0.pc2pc.layer.{i}"] = f"model.0.pc2pc.layer.{i}"
        name_map[f"model.1.pc2pc.layer.{i}"] = f"model.1.pc2pc.layer.{i}"
        name_map[f"model.1.p2p.layer.{i}"] = f"model.1.p2p.layer.{i}"
    name_map["tonic_classifier.3"] = "tonic_map"
    name_map["key_classifier.3"] = "key_map"
    name_map["genre_classifier.3"] = "genre_map"
    save_taps = {}
    for rname, oname in name_map.items():
        # model.1.pool is called once per forward in layer 1 (on pc2), model.0.pool once in layer 0
        check(f"tap {rname}", taps_orc[oname], taps_ref[rname], 1e-12)
        save_taps["tap/" + oname] = taps_ref[rname].numpy().astype(np.float32)
    for a, b, n in zip(outs_o, outs_b, ("key", "tonic", "genre")):
        check(f"B {n}", a, b, 1e-12)
    np.savez_compressed(os.path.join(GOLD, "pcnet_taps_T28.npz"), x=xb.numpy(), seq_length=np.array([28, 28]),
                        key=outs_b[0].numpy(), tonic=outs_b[1].numpy(), genre=outs_b[2].numpy(), **save_taps)

    # ---------------------------------------------------------------- C: 360-bin guard-octave equivariance table
    print("C: equivariance_test.py geometry (octaves=10 -> 360 bins, no genre head), T=40")
    opt10 = default_opt(octaves=10, genre=False)
    net10, sd10 = build_reference_net(opt10, seed=7)
    shifts = _refstubs.reference_functions(os.path.join(_refstubs.REFERENCE_ROOT, "equivariance_test.py"),
                                           ["mel_shifting_up", "mel_shifting_down"])
    mel = (torch.rand((288, 40), generator=g) * 2.5).float().double()
    for s in range(0, 13):      # restated helpers == the reference's loops
        if s:
            check(f"mel_shifting_up({s})", mirex_oracle.mel_shifting_up(mel.numpy(), s), shifts["mel_shifting_up"](mel, s).numpy(), 0)
            check(f"mel_shifting_down({s})", mirex_oracle.mel_shifting_down(mel.numpy(), s), shifts["mel_shifting_down"](mel, s).numpy(), 0)
    # shift_and_stack, equivariance_test.py:172-205: pad a guard octave above and below, 13 up-shifts
    # prepended, 12 down-shifts appended.  We keep tonic too (the script discards it, :188).
    mel_g = torch.cat((torch.zeros(36, 40, dtype=torch.float64), mel, torch.zeros(36, 40, dtype=torch.float64)))
    seq1 = torch.tensor(40).reshape(1, 1)

    def stack(model):
        rows_k, rows_t = [], []
        for i in range(0, 13):
            m = shifts["mel_shifting_up"](mel_g, i) if i > 0 else mel_g
            k, t = model.forward(m.reshape(1, 1, 360, 40), seq1)
            rows_k.insert(0, k[0].detach().numpy()); rows_t.insert(0, t[0].detach().numpy())
        for i in range(1, 13):
            m = shifts["mel_shifting_down"](mel_g, i)
            k, t = model.forward(m.reshape(1, 1, 360, 40), seq1)
            rows_k.append(k[0].detach().numpy()); rows_t.append(t[0].detach().numpy())
        return np.stack(rows_k), np.stack(rows_t)

    net10.eval()
    ek, et = stack(net10)
    net10.train()               # as the script runs it (equivariance_test.py:178: no .eval(), no no_grad)
    sd_before = {k: v.clone() for k, v in net10.state_dict().items()}
    tk, tt = stack(net10)
    net10.load_state_dict(sd_before)
    net10.eval()
    # oracle in eval mode and in train mode (batch statistics)
    ko, to = pcnet_oracle.pcnet_forward(sd10, mel_g.reshape(1, 1, 360, 40), seq1)
    check("C eval key row 12 (unshifted)", ko[0], ek[12], 1e-12)
    check("C eval tonic row 12", to[0], et[12], 1e-12)
    ko, to = pcnet_oracle.pcnet_forward(sd10, mel_g.reshape(1, 1, 360, 40), seq1, training=True)
    check("C train-mode key row 12", ko[0], tk[12], 1e-12)
    # the roll identity of SURVEY section 4.2 on the reference's own outputs
    for s in range(1, 13):
        check(f"reference equivariance up {s}", ek[12 - s], np.roll(ek[12], s), 1e-13)
        check(f"reference equivariance down {s}", ek[12 + s], np.roll(ek[12], -s), 1e-13)
    np.savez_compressed(os.path.join(GOLD, "pcnet_guard360.npz"), opt=json.dumps(vars(opt10)),
                        mel=mel.numpy().astype(np.float32), key_eval=ek, tonic_eval=et, key_train=tk, tonic_train=tt,
                        **sd_to_npz(sd10))

    # ---------------------------------------------------------------- D: MIREX score + table + loss
    print("D: MIREX score, key-signature table, loss")
    table_ref = models.key_sig.KEY_SIGNATURE_MAP.numpy()
    check("KEY_SIGNATURE_MAP", mirex_oracle.key_signature_map(), table_ref, 0)
    rng = np.random.default_rng(5)
    cases = {}
    n = 96
    label_sig = rng.integers(0, 24, n)
    key_labels = np.zeros((n, 12), np.float32)
    for i, s in enumerate(label_sig):   # natural-minor / major pitch-class sets
        tnc = s % 12
        maj_tonic = tnc if s >= 12 else (tnc + 3) % 12
        for st in (0, 2, 4, 5, 7, 9, 11):
            key_labels[i, (maj_tonic + st) % 12] = 1
    tonic_labels = np.eye(12, dtype=np.float32)[label_sig % 12]
    sig_onehot = np.eye(24, dtype=np.float32)[label_sig]
    key_preds = rng.random((n, 12))
    tonic_preds = rng.standard_normal((n, 12))
    # steer a third of the cases to exact / near hits so every category occurs
    for i in range(0, n, 3):
        key_preds[i] = key_labels[i] * 0.8 + 0.1 + rng.random(12) * 0.05
    for i in range(0, n, 2):
        tonic_preds[i, label_sig[i] % 12] += 4.0
    ref_m = net.mirex_score(torch.tensor(key_labels).double(), torch.tensor(key_preds), torch.tensor(tonic_labels).long(),
                            torch.tensor(tonic_preds), torch.tensor(sig_onehot))
    orc_m = mirex_oracle.mirex_score(key_labels, key_preds, tonic_labels, tonic_preds, sig_onehot)
    check("mirex_score 7-tuple", np.array(orc_m), np.array([float(v) for v in ref_m]), 0)
    print("   reference (mirex, correct, fifths, relative, parallel, other, acc) =", [round(float(v), 4) for v in ref_m])
    cases.update(key_labels=key_labels, key_preds=key_preds, tonic_labels=tonic_labels, tonic_preds=tonic_preds,
                 key_signature_id=sig_onehot, mirex=np.array([float(v) for v in ref_m], np.float32), table=table_ref)
    # per-sample categories (batch of one each) so tests can address single cases
    per = np.stack([np.array([float(v) for v in net.mirex_score(
        torch.tensor(key_labels[i:i + 1]).double(), torch.tensor(key_preds[i:i + 1]), torch.tensor(tonic_labels[i:i + 1]).long(),
        torch.tensor(tonic_preds[i:i + 1]), torch.tensor(sig_onehot[i:i + 1]))]) for i in range(n)])
    cases["mirex_per_sample"] = per.astype(np.float32)

    # loss through the reference's own general_step (genre on; one row without a genre label)
    genre_lab = np.eye(11, dtype=np.float32)[rng.integers(0, 11, 4)]
    genre_lab[1] = 0
    batch = {"mel": x.double(), "key_signature_id": torch.tensor(sig_onehot[:4]), "key_labels": torch.tensor(key_labels[:4]),
             "tonic_labels": torch.tensor(tonic_labels[:4]), "genre": torch.tensor(genre_lab), "seq_length": seq}
    out = net.general_step(batch, 0, "val")
    loss_ref = float(out[0])
    loss_orc = loss_oracle.general_step_loss(key.numpy(), tonic.numpy(), genre.numpy(), key_labels[:4], tonic_labels[:4], genre_lab)
    check("general_step loss", loss_orc, loss_ref, 1e-12)
    cases.update(loss_key_labels=key_labels[:4], loss_tonic_labels=tonic_labels[:4], loss_genre_labels=genre_lab,
                 loss_sig=sig_onehot[:4], loss=np.float64(loss_ref),
                 step_metrics=np.array([float(v) for v in out[1:]], np.float32))
    np.savez_compressed(os.path.join(GOLD, "mirex_loss_cases.npz"), **cases)

    # ---------------------------------------------------------------- E: --local (sliding-window key tracking), B=2, T=120
    print("E: --local heads (MaxPool2d((1, 38), stride 1) + per-frame reshape), weights of A, B=2, T=120")
    opt_l = default_opt(local=True)
    torch.manual_seed(0)
    net_l = models.PitchClassNet(opt_l.octaves * 36, 12, opt_l.num_layers, opt_l.kernel_size, opt_l, window_size=opt_l.window_size).double()
    net_l.load_state_dict(sd, strict=True)             # the pooling layers carry no parameters: same keys as the default net
    net_l.eval()
    gl = torch.Generator().manual_seed(321)
    xl = (torch.rand((2, 1, 288, 120), generator=gl) * 2.5).float()
    W = opt_l.frames * opt_l.loc_window_size - opt_l.head_layers * (opt_l.kernel_size - 1)
    kl, tl, gnl = net_l(xl.double(), torch.tensor([120, 120]))
    okl, otl, ogl = pcnet_oracle.pcnet_forward(sd, xl.double(), torch.tensor([120, 120]), local_window=W)
    report["checks"]["E_key"] = check("local key", okl, kl, 1e-12)
    report["checks"]["E_tonic"] = check("local tonic", otl, tl, 1e-12)
    report["checks"]["E_genre"] = check("local genre", ogl, gnl, 1e-12)
    np.savez_compressed(os.path.join(GOLD, "pcnet_local_T120.npz"), x=xl.numpy(), window=np.int64(W),
                        key=kl.numpy(), tonic=tl.numpy(), genre=gnl.numpy())

    # ---------------------------------------------------------------- F: --resblock, B=2, T=28
    print("F: --resblock (ResBlock / ResBlockEquivariant stacks), seeded weights, B=2, T=28")
    opt_r = default_opt(resblock=True)
    net_r, sd_r = build_reference_net(opt_r, seed=21)
    net_r.eval()
    gr = torch.Generator().manual_seed(654)
    xr = (torch.rand((2, 1, 288, 28), generator=gr) * 2.5).float()
    seq_r = torch.tensor([28, 22])
    kr, tr, gnr = net_r(xr.double(), seq_r)
    okr, otr, ogr = pcnet_oracle.pcnet_forward(sd_r, xr.double(), seq_r)
    report["checks"]["F_key"] = check("resblock key", okr, kr, 1e-12)
    report["checks"]["F_tonic"] = check("resblock tonic", otr, tr, 1e-12)
    report["checks"]["F_genre"] = check("resblock genre", ogr, gnr, 1e-12)
    np.savez_compressed(os.path.join(GOLD, "pcnet_resblock_T28.npz"), opt=json.dumps(vars(opt_r)), x=xr.numpy(), seq_length=seq_r.numpy(),
                        key=kr.numpy(), tonic=tr.numpy(), genre=gnr.numpy(), **sd_to_npz(sd_r))

    # ---------------------------------------------------------------- G: --pc2p_mem, B=2, T=40
    print("G: --pc2p_mem (PitchClass2Pitch_MemoryVariant: the up_sixth map is summed over its 4 channels and ADDED to the pitch stream), B=2, T=40")
    opt_m = default_opt(pc2p_mem=True)
    net_m, sd_m = build_reference_net(opt_m, seed=33)
    net_m.eval()
    gm = torch.Generator().manual_seed(987)
    xm = (torch.rand((2, 1, 288, 40), generator=gm) * 2.5).float()
    seq_m = torch.tensor([40, 31])
    km, tm, gnm = net_m(xm.double(), seq_m)
    okm, otm, ogm = pcnet_oracle.pcnet_forward(sd_m, xm.double(), seq_m)
    report["checks"]["G_key"] = check("pc2p_mem key", okm, km, 1e-12)
    report["checks"]["G_tonic"] = check("pc2p_mem tonic", otm, tm, 1e-12)
    report["checks"]["G_genre"] = check("pc2p_mem genre", ogm, gnm, 1e-12)
    np.savez_compressed(os.path.join(GOLD, "pcnet_pc2pmem_T40.npz"), opt=json.dumps(vars(opt_m)), x=xm.numpy(), seq_length=seq_m.numpy(),
                        key=km.numpy(), tonic=tm.numpy(), genre=gnm.numpy(), **sd_to_npz(sd_m))

    # ---------------------------------------------------------------- H: --p2pc_conv, B=2, T=40
    print("H: --p2pc_conv (Pitch2PitchClassConv: the octave fold as a dilated conv + BN + LeakyReLU), B=2, T=40")
    opt_c = default_opt(p2pc_conv=True)
    net_c, sd_c = build_reference_net(opt_c, seed=44)
    net_c.eval()
    gc = torch.Generator().manual_seed(246)
    xc = (torch.rand((2, 1, 288, 40), generator=gc) * 2.5).float()
    seq_c = torch.tensor([40, 33])
    kc, tc, gnc = net_c(xc.double(), seq_c)
    okc, otc, ogc = pcnet_oracle.pcnet_forward(sd_c, xc.double(), seq_c)
    report["checks"]["H_key"] = check("p2pc_conv key", okc, kc, 1e-12)
    report["checks"]["H_tonic"] = check("p2pc_conv tonic", otc, tc, 1e-12)
    report["checks"]["H_genre"] = check("p2pc_conv genre", ogc, gnc, 1e-12)
    np.savez_compressed(os.path.join(GOLD, "pcnet_p2pcconv_T40.npz"), opt=json.dumps(vars(opt_c)), x=xc.numpy(), seq_length=seq_c.numpy(),
                        key=kc.numpy(), tonic=tc.numpy(), genre=gnc.numpy(), **sd_to_npz(sd_c))

    # ---------------------------------------------------------------- I: --stay_sixth, three layers, B=2, T=64
    print("I: --stay_sixth (the pitch stream stays at semitone resolution: 96 rows, no up_sixth / pool_semi after layer 0), num_layers=2, B=2, T=40")
    opt_s = default_opt(stay_sixth=True)
    net_s, sd_s = build_reference_net(opt_s, seed=55)
    net_s.eval()
    gs = torch.Generator().manual_seed(135)
    xs = (torch.rand((2, 1, 288, 40), generator=gs) * 2.5).float()
    seq_s = torch.tensor([40, 29])
    ks_, ts_, gns_ = net_s(xs.double(), seq_s)
    oks, ots, ogs = pcnet_oracle.pcnet_forward(sd_s, xs.double(), seq_s)
    report["checks"]["I_key"] = check("stay_sixth key", oks, ks_, 1e-12)
    report["checks"]["I_tonic"] = check("stay_sixth tonic", ots, ts_, 1e-12)
    report["checks"]["I_genre"] = check("stay_sixth genre", ogs, gns_, 1e-12)
    np.savez_compressed(os.path.join(GOLD, "pcnet_staysixth_T40.npz"), opt=json.dumps(vars(opt_s)), x=xs.numpy(), seq_length=seq_s.numpy(),
                        key=ks_.numpy(), tonic=ts_.numpy(), genre=gns_.numpy(), **sd_to_npz(sd_s))

    # ---------------------------------------------------------------- J: --denseblock, B=2, T=40
    print("J: --denseblock (DenseNet-style stacks: pre-activation BatchNorm, 1-wide bottlenecks, concatenated features), n_filters=2, conv_layers=2, B=2, T=40")
    opt_d = default_opt(denseblock=True, n_filters=2, conv_layers=2)
    net_d, sd_d = build_reference_net(opt_d, seed=66)
    net_d.eval()
    gd = torch.Generator().manual_seed(146)
    xd = (torch.rand((2, 1, 288, 40), generator=gd) * 2.5).float()
    seq_d = torch.tensor([40, 31])
    kd_, td_, gnd_ = net_d(xd.double(), seq_d)
    okd, otd, ogd = pcnet_oracle.pcnet_forward(sd_d, xd.double(), seq_d)
    report["checks"]["J_key"] = check("denseblock key", okd, kd_, 1e-12)
    report["checks"]["J_tonic"] = check("denseblock tonic", otd, td_, 1e-12)
    report["checks"]["J_genre"] = check("denseblock genre", ogd, gnd_, 1e-12)
    # the default widths (n_filters=4, conv_layers=3: 76-channel bottleneck, 51 -> 102 channel heads) are checked here but not stored (4 MB)
    opt_d2 = default_opt(denseblock=True)
    net_d2, sd_d2 = build_reference_net(opt_d2, seed=67)
    net_d2.eval()
    kd2, td2, gnd2 = net_d2(xd.double(), seq_d)
    okd2, otd2, ogd2 = pcnet_oracle.pcnet_forward(sd_d2, xd.double(), seq_d)
    report["checks"]["J_default_widths_key"] = check("denseblock (default widths) key", okd2, kd2, 1e-12)
    report["checks"]["J_default_widths_tonic"] = check("denseblock (default widths) tonic", otd2, td2, 1e-12)
    report["checks"]["J_default_widths_genre"] = check("denseblock (default widths) genre", ogd2, gnd2, 1e-12)
    np.savez_compressed(os.path.join(GOLD, "pcnet_denseblock_T40.npz"), opt=json.dumps(vars(opt_d)), x=xd.numpy(), seq_length=seq_d.numpy(),
                        key=kd_.numpy(), tonic=td_.numpy(), genre=gnd_.numpy(), **sd_to_npz(sd_d))

    # ---------------------------------------------------------------- K: --kernel_size 3 and 5 (train_model.py:194), B=2, T=40
    for ksz, seed_k in ((3, 77), (5, 78)):
        print(f"K: --kernel_size {ksz} (k x k pitch convs, 12 x k pitch-class convs and heads, (2, k) genre conv; heads shrink by k - 1 per layer), B=2, T=40")
        opt_k = default_opt(kernel_size=ksz)
        net_k, sd_k = build_reference_net(opt_k, seed=seed_k)
        net_k.eval()
        gk = torch.Generator().manual_seed(150 + ksz)
        xk = (torch.rand((2, 1, 288, 40), generator=gk) * 2.5).float()
        seq_k = torch.tensor([40, 27])
        kk_, tk_, gnk_ = net_k(xk.double(), seq_k)
        okk, otk, ogk = pcnet_oracle.pcnet_forward(sd_k, xk.double(), seq_k, kernel_size=ksz)
        report["checks"][f"K{ksz}_key"] = check(f"kernel_size {ksz} key", okk, kk_, 1e-12)
        report["checks"][f"K{ksz}_tonic"] = check(f"kernel_size {ksz} tonic", otk, tk_, 1e-12)
        report["checks"][f"K{ksz}_genre"] = check(f"kernel_size {ksz} genre", ogk, gnk_, 1e-12)
        kn_, tn_, gnn_ = net_k(xk.double(), None)
        okn, otn, ogn = pcnet_oracle.pcnet_forward(sd_k, xk.double(), None, kernel_size=ksz)
        report["checks"][f"K{ksz}_key_noseq"] = check(f"kernel_size {ksz} key (no seq_length)", okn, kn_, 1e-12)
        np.savez_compressed(os.path.join(GOLD, f"pcnet_k{ksz}_T40.npz"), opt=json.dumps(vars(opt_k)), x=xk.numpy(), seq_length=seq_k.numpy(),
                            key=kk_.numpy(), tonic=tk_.numpy(), genre=gnk_.numpy(), key_noseq=kn_.numpy(), tonic_noseq=tn_.numpy(),
                            genre_noseq=gnn_.numpy(), **sd_to_npz(sd_k))

    # ---------------------------------------------------------------- L: --denseblock, ONE training forward + backward through the real reference
    print("L: --denseblock in train() mode: loss, every parameter's gradient (the reference's own autograd, float64) and the BatchNorm running "
          "statistics after the step (the dense layers checkpoint norm1 + conv1, models.py:484-489, 510-514: that half runs twice per step)")
    with torch.enable_grad():
        opt_t = default_opt(denseblock=True, n_filters=2, conv_layers=2)
        net_t, sd_t = build_reference_net(opt_t, seed=88)
        sd_t = {k: v.clone() for k, v in sd_t.items()}        # (the integer buffers of state_dict() are the module's own: the step below counts them up)
        net_t.train()
        gt = torch.Generator().manual_seed(188)
        xt = (torch.rand((2, 1, 288, 40), generator=gt) * 2.5).float()
        seq_t = torch.tensor([40, 33])
        key_labels = (torch.rand((2, 12), generator=gt) > 0.5).double()
        tonic_idx = torch.randint(0, 12, (2,), generator=gt)
        genre_idx = torch.randint(0, 11, (2,), generator=gt)
        import torch.nn.functional as F_

        def loss_of(out):      # models.py:878-893 with the default weights (key 1, tonic 1, genre 0.1), every clip labelled
            return F_.binary_cross_entropy(out[0], key_labels) + F_.cross_entropy(out[1], tonic_idx) + 0.1 * F_.cross_entropy(out[2], genre_idx)

        loss_t = loss_of(net_t(xt.double(), seq_t))
        loss_t.backward()
        grads_t = {k: p.grad.detach().clone() for k, p in net_t.named_parameters()}
        after_t = {k: v.detach().clone() for k, v in net_t.state_dict().items() if "running_" in k}
        # the restatement's autograd against the reference's
        sd_o = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd_t.items()}
        loss_o = loss_of(pcnet_oracle.pcnet_forward(sd_o, xt.double(), seq_t, training=True))
        loss_o.backward()
        report["checks"]["L_loss"] = check("denseblock train-mode loss", loss_o.detach(), loss_t.detach(), 1e-12)
        # (biases in front of a BatchNorm have an exactly-zero gradient: 1e-18 against 1e-18 -- the error is taken relative to at least 1e-9 of
        # the step's largest gradient)
        gmax = max(float(g.abs().max()) for g in grads_t.values())
        worst = max(float((sd_o[k].grad - g).abs().max() / max(float(g.abs().max()), 1e-9 * gmax)) for k, g in grads_t.items())
        print(f"  oracle autograd vs reference autograd, worst tensor (relative to its max): {worst:.3e}")
        assert worst < 1e-6, worst
        report["checks"]["L_grad_worst_rel"] = worst
        # which BatchNorm layers saw the batch twice?  running = (1 - m)^r * old + (1 - (1 - m)^r) * batch mean, m = 0.1
        twice = []
        for k in sorted(after_t):
            if not k.endswith("running_mean"):
                continue
            old, new = sd_t[k], after_t[k]
            bn_in = None
            twice.append((k, new, old))
        np.savez_compressed(os.path.join(GOLD, "pcnet_denseblock_train_T40.npz"), opt=json.dumps(vars(opt_t)), x=xt.numpy(), seq_length=seq_t.numpy(),
                            key_labels=key_labels.numpy(), tonic_idx=tonic_idx.numpy(), genre_idx=genre_idx.numpy(), loss=loss_t.detach().numpy(),
                            **{"grad/" + k: v.numpy() for k, v in grads_t.items()}, **{"after/" + k: v.numpy() for k, v in after_t.items()},
                            **sd_to_npz(sd_t))
    torch.set_grad_enabled(False)

    with open(os.path.join(GOLD, "PROVENANCE.json"), "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", sorted(os.listdir(GOLD)))


if __name__ == "__main__":
    main()
