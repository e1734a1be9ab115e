"""CPU tests of the boundary: the shared library loads, exports everything include/asrfe.h declares,
its host-side tables agree with the oracle, and the sharding helpers behave (incl. a 2-rank gloo run).
No compute entry point is called here (there is no GPU)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from oracle import fbank_oracle as fo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    import toolbox_for_asr_and_tts_amd as p

    return p


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "asrfe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(asrfe_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load()
    names = _declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/asrfe.h but not exported"
    assert set(names) == set(pkg._lib.SIGNATURES), "ctypes SIGNATURES out of sync with the header"
    assert lib.asrfe_version().startswith(b"asrfe")


def test_default_config_is_paraformer_large(pkg):
    c = pkg.default_config()
    assert (c.sample_rate, c.frame_length, c.frame_shift, c.n_fft, c.n_mels) == (16000, 400, 160, 512, 80)
    assert (c.lfr_m, c.lfr_n) == (7, 6)
    assert (c.enc_input_size, c.enc_output_size, c.enc_heads, c.enc_linear_units, c.enc_num_blocks) == (560, 512, 4, 2048, 50)
    assert c.enc_kernel_size == 11 and c.enc_sanm_shift == 0
    assert c.enc_compute_dtype == pkg._lib.DT_F16
    assert abs(c.preemph - 0.97) < 1e-7 and abs(c.enc_ln_eps - 1e-12) < 1e-18


def test_frame_count_helpers_match_oracle(pkg):
    lib = pkg.load()
    for n in [0, 1, 399, 400, 401, 559, 560, 1367, 16000, 48000, 480000, 480159, 480160]:
        assert lib.asrfe_num_frames(None, n) == fo.num_frames(n)
    for t in [0, 1, 5, 6, 7, 298, 2998]:
        assert lib.asrfe_num_lfr_frames(None, t) == fo.num_lfr_frames(t)


def test_host_tables_match_oracle(pkg):
    lib = pkg.load()
    cfg = pkg.default_config()
    banks = np.zeros((80, 256), dtype=np.float64)
    assert lib.asrfe_host_mel_banks(C.byref(cfg), banks.ctypes.data_as(C.POINTER(C.c_double))) == 0
    assert np.max(np.abs(banks - fo.mel_banks()[:, :256])) < 1e-12
    win = np.zeros(400, dtype=np.float64)
    assert lib.asrfe_host_window(C.byref(cfg), win.ctypes.data_as(C.POINTER(C.c_double))) == 0
    assert np.max(np.abs(win - fo.hamming_window())) < 1e-15


def test_shard_range_partitions(pkg):
    for n, w in [(512, 8), (64, 1), (10, 4), (3, 8), (0, 2)]:
        ranges = [pkg.shard_range(n, r, w) for r in range(w)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        for (a, b), (c, d) in zip(ranges, ranges[1:]):
            assert b == c
        sizes = [b - a for a, b in ranges]
        assert max(sizes) - min(sizes) <= 1
    assert pkg.shard_range(512, 3, 8) == (192, 256)
    with pytest.raises(ValueError):
        pkg.shard_range(4, 4, 4)


def test_shard_by_length_balances_ragged_batches(pkg):
    rng = np.random.default_rng(3)
    lens = rng.integers(16000 * 5, 16000 * 30, 512).tolist()
    for w in (1, 2, 4, 8):
        parts = [pkg.shard_by_length(lens, r, w) for r in range(w)]
        assert sorted(i for p in parts for i in p) == list(range(512))      # a partition
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
        audio = [sum(lens[i] for i in p) for p in parts]
        assert max(audio) - min(audio) <= 0.01 * max(audio)                 # within 1 % of each other
        for p in parts:                                                     # each rank's clips come sorted by length
            assert [lens[i] for i in p] == sorted(lens[i] for i in p)
    assert pkg.shard_by_length([], 0, 2) == []
    with pytest.raises(ValueError):
        pkg.shard_by_length(lens, 2, 2)


def test_missing_library_fails_loudly(pkg, monkeypatch):
    monkeypatch.setattr(pkg._lib, "_lib", None)
    monkeypatch.setattr(pkg._lib, "LIB_PATH", "/nonexistent/libasrfe.so")
    with pytest.raises(ImportError):
        pkg._lib.load()


def test_compute_entry_points_refuse_host_tensors(pkg):
    import torch

    # constructing a frontend needs a device; without one the library must raise, not fall back
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tests")
    with pytest.raises(pkg.AsrfeError):
        pkg.WavFrontend()


def test_two_rank_gloo_sharding():
    """N > 1 path on CPU: two gloo ranks shard 6 utterances, each runs the oracle on its slice, the
    gathered frame counts and checksums equal the single-process result and timing is max-reduced."""
    script = os.path.join(ROOT, "tests", "gloo_shard_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", script],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "SHARD_OK" in r.stdout


def test_eight_rank_gloo_plans():
    """The node's full width on CPU: eight gloo ranks divide the batch the three ways bench.py and AsrEngine do
    (weak 64 per rank, strong 512 split, ragged 512 dealt out by length); each is a partition, the ragged one is
    balanced to within one clip of audio, and sampled clips give the single-process oracle's features."""
    script = os.path.join(ROOT, "tests", "gloo_shard_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", script, "plans"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PLANS_OK 8" in r.stdout
