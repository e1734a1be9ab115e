"""``python bench.py --gpus 2`` without a launcher starts two ranks itself.  On the one-GPU test box both ranks share
cuda:0 (ASRFE_BENCH_REHEARSAL=1: gloo instead of RCCL, which refuses two ranks on one device), so the numbers mean
nothing; what is checked is the protocol: two ranks took part, the line says so, and the units are those of both."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(280)
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_plain_gpus_2_call_runs_two_ranks(scaling):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ASRFE_BENCH_REHEARSAL"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
           "--no-cpu-baseline", "--no-extras", "--no-stages", "--scaling", scaling, "--global-batch", "6"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=270, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == scaling
    assert d["config"]["global_batch"] == (8 if scaling == "weak" else 6)
    assert d["output_finite"] is True
    assert abs(d["value"] - d["config"]["global_batch"] * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) < 1e-6 * d["value"]


@pytest.mark.timeout(280)
def test_one_rank_through_the_rccl_path():
    """The barrier and the reductions of a multi-GPU run go through RCCL; a box with one GPU cannot host two RCCL ranks, so
    one rank forms the group alone (ASRFE_BENCH_FORCE_DIST=1): process-group set-up on the device, barrier, MAX / SUM
    all-reduces of device tensors, tear-down."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ASRFE_BENCH_FORCE_DIST"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4",
           "--no-cpu-baseline", "--no-extras", "--no-stages"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=270, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["collective_backend"] == "nccl"
    assert d["config"]["global_batch"] == 4 and d["output_finite"] is True
