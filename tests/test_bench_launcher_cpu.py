"""bench.py's multi-rank entry, as far as it can be exercised without a GPU: a plain ``--gpus N`` call must start
ranks itself or fail loudly, never measure one GPU and report it as N."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                             "ASRFE_BENCH_REHEARSAL")}
    env.update(extra)
    return env


def test_more_gpus_than_visible_is_an_error():
    n = torch.cuda.device_count() + 1 if torch.cuda.device_count() else 2
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert r.returncode != 0
    assert f"--gpus {n} but only" in r.stderr
    assert '"n_gpus"' not in r.stdout  # no measurement line


def test_world_size_must_equal_gpus():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1"], capture_output=True, text=True, timeout=300,
                       env=_clean_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], capture_output=True, text=True, timeout=300,
                       env=_clean_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


@pytest.mark.skipif(torch.cuda.is_available(), reason="covers the box without a GPU")
def test_no_gpu_is_an_error_not_a_cpu_run():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "1"], capture_output=True, text=True,
                       timeout=300, env=_clean_env(), cwd=ROOT)
    assert r.returncode != 0 and "needs a GPU" in r.stderr
