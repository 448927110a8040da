"""The C-ABI library loads without a GPU and exports exactly what include/ake_hip.h declares."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import ake_amd
from ake_amd import _lib
from conftest import REPO


def header_symbols():
    src = open(os.path.join(REPO, "include", "ake_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(ake_[a-z0-9_]+)\s*\(", src))


def test_every_declared_symbol_is_exported_and_bound():
    declared = header_symbols()
    assert len(declared) >= 25
    lib = _lib.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in ake_hip.h but not exported by libake_hip.so"
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))


def test_version_and_error_string():
    lib = _lib.lib()
    assert lib.ake_version() >= 100
    assert isinstance(lib.ake_last_error(), bytes)


def test_default_configs_follow_the_reference_flags():
    lib = _lib.lib()
    c = _lib.CqtConfig()
    assert lib.ake_cqt_default_config(C.byref(c), 22050, 5, 8) == 0
    assert (c.sample_rate, c.hop_length, c.n_bins, c.bins_per_octave) == (22050, 4410, 288, 36)   # KeyDataset.py:485,491
    assert lib.ake_cqt_default_config(C.byref(c), 44100, 5, 8) == 0 and c.hop_length == 8820
    assert lib.ake_cqt_default_config(C.byref(c), 22050, 4, 8) == 0 and c.hop_length == round(22050 / 4)   # 5512 (half-even)
    p = _lib.PcnetConfig()
    assert lib.ake_pcnet_default_config(C.byref(p), 8, 1) == 0
    assert (p.pitches, p.num_layers, p.kernel_size, p.conv_layers, p.n_filters, p.head_layers, p.time_pool_size, p.genre) == \
        (288, 2, 7, 3, 4, 2, 2, 1)                                                                 # train_model.py:190-217


def _create(cfg_kw=None, octaves=8, genre=1):
    lib = _lib.lib()
    p = _lib.PcnetConfig()
    lib.ake_pcnet_default_config(C.byref(p), octaves, genre)
    for k, v in (cfg_kw or {}).items():
        setattr(p, k, v)
    h = C.c_void_p()
    rc = lib.ake_pcnet_create(C.byref(p), C.byref(h))
    return rc, h


def test_tensor_registry_equals_reference_state_dict(gold_default, gold_guard):
    """Names and shapes the handle expects == float entries of the reference state_dict (strict=True contract)."""
    lib = _lib.lib()
    for gold, octaves, genre in ((gold_default, 8, 1), (gold_guard, 10, 0)):
        rc, h = _create(octaves=octaves, genre=genre)
        assert rc == 0
        want = {k[3:]: gold[k].shape for k in gold.files if k.startswith("sd/") and not k.endswith("num_batches_tracked")}
        got = {}
        for i in range(lib.ake_pcnet_num_tensors(h)):
            name, shape, nd = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
            assert lib.ake_pcnet_tensor_info(h, i, C.byref(name), shape, C.byref(nd)) == 0
            got[name.value.decode()] = tuple(shape[:nd.value])
        assert got == want
        lib.ake_pcnet_destroy(h)


def test_set_tensor_validation_and_strict_finalize(gold_default):
    lib = _lib.lib()
    rc, h = _create()
    w = np.ascontiguousarray(gold_default["sd/model.1.p2p.layer.0.weight"])
    shape = (C.c_int64 * 4)(*w.shape)
    assert lib.ake_pcnet_set_tensor(h, b"model.1.p2p.layer.0.weight", w.ctypes.data, shape, 4) == 0
    bad = (C.c_int64 * 4)(8, 5, 7, 5)
    assert lib.ake_pcnet_set_tensor(h, b"model.1.p2p.layer.0.weight", w.ctypes.data, bad, 4) == -1
    assert b"dim 3" in lib.ake_last_error()
    assert lib.ake_pcnet_set_tensor(h, b"model.7.nope", w.ctypes.data, shape, 4) == -1
    assert lib.ake_pcnet_finalize(h) == -3                       # AKE_ERR_STATE: keys missing (strict)
    assert b"missing key" in lib.ake_last_error()
    lib.ake_pcnet_destroy(h)


@pytest.mark.parametrize("flag", ["only_semitones"])
def test_variant_flags_are_refused(flag):
    rc, h = _create({flag: 1})
    assert rc == -5 and b"not built" in _lib.lib().ake_last_error()


@pytest.mark.parametrize("flag", ["resblock", "pc2p_mem", "p2pc_conv", "stay_sixth"])
def test_variant_tensor_registry_equals_reference_state_dict(flag, gold_resblock, gold_pc2pmem, gold_p2pcconv, gold_staysixth):
    """cfg.resblock / cfg.pc2p_mem = 1: the handle expects exactly the float entries of the reference's state_dict for that flag."""
    lib = _lib.lib()
    rc, h = _create({flag: 1})
    assert rc == 0
    gold = {"resblock": gold_resblock, "pc2p_mem": gold_pc2pmem, "p2pc_conv": gold_p2pcconv, "stay_sixth": gold_staysixth}[flag]
    want = {k[3:]: tuple(gold[k].shape) for k in gold.files if k.startswith("sd/") and gold[k].dtype.kind == "f"}
    got = {}
    for i in range(lib.ake_pcnet_num_tensors(h)):
        name, shape, ndim = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
        assert lib.ake_pcnet_tensor_info(h, i, C.byref(name), shape, C.byref(ndim)) == 0
        got[name.value.decode()] = tuple(shape[d] for d in range(ndim.value))
    assert got == want
    lib.ake_pcnet_destroy(h)


def test_denseblock_tensor_registry_equals_reference_state_dict(gold_denseblock):
    """cfg.denseblock = 1 (with the fixture's n_filters / conv_layers): the handle expects exactly the float entries of the reference's
    state_dict; combinations with the other variants are refused."""
    lib = _lib.lib()
    rc, h = _create({"denseblock": 1, "n_filters": 2, "conv_layers": 2})
    assert rc == 0
    want = {k[3:]: tuple(gold_denseblock[k].shape) for k in gold_denseblock.files if k.startswith("sd/") and gold_denseblock[k].dtype.kind == "f"}
    got = {}
    for i in range(lib.ake_pcnet_num_tensors(h)):
        name, shape, ndim = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
        assert lib.ake_pcnet_tensor_info(h, i, C.byref(name), shape, C.byref(ndim)) == 0
        got[name.value.decode()] = tuple(shape[d] for d in range(ndim.value))
    assert got == want
    lib.ake_pcnet_destroy(h)
    rc, _ = _create({"denseblock": 1, "resblock": 1})
    assert rc == -5 and b"not built" in lib.ake_last_error()


def test_local_config_is_the_pooling_window():
    """cfg.local = W > 0 builds the --local net (same tensors as the default net; models.py:720-722 adds parameter-free pooling):
    frame arithmetic without time pooling, and the clip-level entry point is refused."""
    lib = _lib.lib()
    rc, h = _create({"local": 38})
    assert rc == 0
    rc0, h0 = _create()
    assert lib.ake_pcnet_num_tensors(h) == lib.ake_pcnet_num_tensors(h0)
    tq, tm = C.c_int(), C.c_int()
    assert lib.ake_pcnet_local_frames(h, 120, C.byref(tq), C.byref(tm)) == 0 and (tq.value, tm.value) == (71, 108)
    assert lib.ake_pcnet_local_frames(h0, 120, C.byref(tq), C.byref(tm)) == -3          # AKE_ERR_STATE: not a --local net
    rc, _ = _create({"local": -1})
    assert rc == -1
    lib.ake_pcnet_destroy(h); lib.ake_pcnet_destroy(h0)


def test_workspace_queries_need_no_gpu():
    lib = _lib.lib()
    rc, h = _create()
    small, big = lib.ake_pcnet_workspace_bytes(h, 1, 76), lib.ake_pcnet_workspace_bytes(h, 256, 76)
    assert 0 < small < big
    huge = lib.ake_pcnet_workspace_bytes(h, 1024, 76)          # pitch stream chunked at 256 clips; only the small pitch-class tail buffers
    assert big < huge < 2.1 * big                               # and layer 0's f16 copy of the log-CQT (88 KB per clip) grow with the batch
    lib.ake_pcnet_destroy(h)


def test_shipped_library_reads_no_environment_switch():
    """the shipped build has no switch an inherited environment variable could flip -- the AKE_* knobs of the kernel
    experiments exist only under -DAKE_DIAG (`AKE_DIAG=1 csrc/build.sh` -> libake_hip_diag.so).  The library does not even import getenv."""
    import subprocess
    lib = _lib.lib()
    assert lib.ake_build_has_diag() == 0
    assert os.path.basename(_lib.LIB_PATH) == "libake_hip.so"
    syms = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True)
    if syms.returncode == 0:                                        # (binutils present on the build and the GPU image)
        assert "getenv" not in syms.stdout


def test_precision_is_part_of_the_config():
    """ake_pcnet_config::precision: MIXED (default) / F32X3, anything else refused; the handle reports what it runs."""
    lib = _lib.lib()
    p = _lib.PcnetConfig()
    lib.ake_pcnet_default_config(C.byref(p), 8, 1)
    assert p.precision == 0
    for prec, want in ((0, 0), (1, 0), (2, -1), (-1, -1)):
        rc, h = _create({"precision": prec})
        assert rc == want, (prec, rc)
        if rc == 0:
            assert lib.ake_pcnet_precision(h) == prec
            lib.ake_pcnet_destroy(h)
    from argparse import Namespace
    assert ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(genre=True)).precision == 0
    assert ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(genre=True, precision="f32x3")).precision == 1
    with pytest.raises(ValueError):
        ake_amd.PitchClassNet(288, 12, 2, 7, Namespace(genre=True, precision="fp8"))
