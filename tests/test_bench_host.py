"""bench.py's host logic that needs no GPU: the launcher, the world-size check, the PMC-profile freshness check."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

BENCH = os.path.join(REPO, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


@pytest.mark.parametrize("extra", [[], ["--train"]])
def test_world_size_must_equal_gpus(extra):
    """VERDICT r1 / ADVICE: `--gpus 8` in a one-rank world used to print an n_gpus: 1 line.  Now: non-zero exit, no line."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--no-cpu-baseline"] + extra,
                         env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "WORLD_SIZE=1 but --gpus 2" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_gpus_n_without_launcher_spawns_n_ranks():
    """With WORLD_SIZE unset the parent starts torch.distributed.run itself (before any GPU call) and relays the ranks' exit code.
    There is no GPU here, so both ranks stop at 'needs a GPU' -- what is checked is that TWO ranks were started by the parent and
    that their failure is the parent's failure (no silent single-rank line)."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                         env=_env(AKE_REHEARSE_ONE_GPU="1"), capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert "2-rank launch failed" in out.stderr
    assert out.stderr.count("bench.py needs a GPU") >= 2 or ("rank: 1" in out.stderr or "local_rank: 1" in out.stderr)
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_pmc_traffic_is_refused_when_stale(tmp_path, monkeypatch):
    sys.path.insert(0, REPO)
    import bench
    h = bench.kernel_set_hash()
    assert len(h) == 16 and h == bench.kernel_set_hash()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "REPO", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_set_hash", lambda: h)
    (prof / "r09_a_pmc_traffic.json").write_text(json.dumps({"kernel_set": "0123456789abcdef", "kernels": {"k": {"hbm_bytes": 1}}}))
    name, kernels, why = bench.pmc_traffic()
    assert kernels == {} and "STALE" in why and name == "r09_a_pmc_traffic.json"
    (prof / "r09_b_pmc_traffic.json").write_text(json.dumps({"kernel_set": h, "kernels": {"k": {"hbm_bytes": 1}}}))
    name, kernels, why = bench.pmc_traffic()
    assert kernels == {"k": {"hbm_bytes": 1}} and why is None and name == "r09_b_pmc_traffic.json"
