import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the float64 oracle runs on the host: a GPU box reports 128+ hardware threads but grants this job a 16-core share, and torch's default
    # (one thread per reported core) oversubscribes it (measured: the config-3 oracle loop 420 s with 128 threads)
    import torch
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


def load_golden(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def golden_state_dict(npz, dtype=None):
    import torch
    sd = {}
    for k in npz.files:
        if k.startswith("sd/"):
            t = torch.from_numpy(npz[k])
            if dtype is not None and t.is_floating_point():
                t = t.to(dtype)
            sd[k[3:]] = t
    return sd


@pytest.fixture(scope="session")
def gold_default():
    return load_golden("pcnet_default.npz")


@pytest.fixture(scope="session")
def gold_taps():
    return load_golden("pcnet_taps_T28.npz")


@pytest.fixture(scope="session")
def gold_guard():
    return load_golden("pcnet_guard360.npz")


@pytest.fixture(scope="session")
def gold_local():
    return load_golden("pcnet_local_T120.npz")


@pytest.fixture(scope="session")
def gold_resblock():
    return load_golden("pcnet_resblock_T28.npz")


@pytest.fixture(scope="session")
def gold_pc2pmem():
    return load_golden("pcnet_pc2pmem_T40.npz")


@pytest.fixture(scope="session")
def gold_p2pcconv():
    return load_golden("pcnet_p2pcconv_T40.npz")


@pytest.fixture(scope="session")
def gold_staysixth():
    return load_golden("pcnet_staysixth_T40.npz")


@pytest.fixture(scope="session")
def gold_denseblock():
    return load_golden("pcnet_denseblock_T40.npz")


@pytest.fixture(scope="session")
def gold_mirex():
    return load_golden("mirex_loss_cases.npz")


def rel_err(a, b):
    """max|a-b| / max(|b|, 1e-6) -- the per-tensor measure of SURVEY.md section 8d."""
    a, b = (t.detach().cpu().numpy() if hasattr(t, "detach") else t for t in (a, b))
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-6))
