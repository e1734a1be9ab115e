"""The C++ CPU restatement of the frontend + encoder entry points (oracle/cpu_abi, test infrastructure) against the
golden fixtures and the numpy / torch oracles: a second, independently written checker with the device library's own
entry points, and what bench.py times as the CPU baseline."""
import glob
import os
import re

import numpy as np
import pytest
import torch

from oracle import fbank_oracle as fo
from oracle import sanm_oracle as so
from tests.cpu_abi_util import CpuModel, load

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_the_frontend_and_encoder_symbols_of_the_header():
    lib = load()
    hdr = open(os.path.join(ROOT, "include", "asrfe.h")).read()
    declared = set(re.findall(r"\b(asrfe_[a-z_0-9]+)\s*\(", hdr))
    subset = {"asrfe_default_config", "asrfe_create", "asrfe_destroy", "asrfe_load_cmvn", "asrfe_load_encoder_weights",
              "asrfe_num_frames", "asrfe_num_lfr_frames", "asrfe_fbank", "asrfe_lfr_cmvn", "asrfe_frontend",
              "asrfe_encoder_workspace_bytes", "asrfe_encoder_forward", "asrfe_last_error", "asrfe_version"}
    assert subset <= declared
    for name in subset:
        assert hasattr(lib, name), name
    assert b"test infrastructure" in lib.asrfe_version()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "fbank_*.npz"))))
def test_frontend_fixtures(path):
    g = np.load(path)
    pcm = g["pcm"][None, :]
    m = CpuModel(num_blocks=1, cmvn=(g["shift"], g["scale"]))
    logmel, nf = m.fbank(pcm, [pcm.shape[1]])
    assert nf.tolist() == [g["logmel"].shape[0]]
    assert np.max(np.abs(logmel[0] - g["logmel"])) <= 1e-3 if nf[0] else True
    feats, ol = m.frontend(pcm, [pcm.shape[1]])
    assert ol.tolist() == [g["feats"].shape[0]]
    if ol[0]:
        assert np.max(np.abs(feats[0] - g["feats"])) <= 1e-3
    # float32 PCM (int16 / 32768) is the same signal
    f2, _ = m.frontend((pcm / 32768.0).astype(np.float32), [pcm.shape[1]])
    assert np.array_equal(f2, feats)
    m.close()


def test_frontend_ragged_batch_against_the_numpy_oracle():
    lens = [0, 399, 400, 560, 4801, 16000]
    pcm = np.zeros((len(lens), 16000), dtype=np.int16)
    for i, n in enumerate(lens):
        pcm[i, :n] = fo.synth_pcm("gauss" if i % 2 else "bursts", n, 30 + i)
    want, wl = fo.frontend(pcm, lens)
    m = CpuModel(num_blocks=1)
    got, gl = m.frontend(pcm, lens)
    assert gl.tolist() == wl.tolist() == [0, 0, 1, 1, 5, 17]
    assert got.shape == want.shape and np.max(np.abs(got - want)) <= 1e-3
    for b, n in enumerate(gl):
        assert not got[b, n:].any()
    # truncation to fewer output rows than the clip has
    cut, cl = m.frontend(pcm, lens, tp_max=3)
    assert cl.tolist() == [0, 0, 1, 1, 3, 3] and np.array_equal(cut[4, :2], got[4, :2])
    m.close()


def test_encoder_tiny_fixture_and_threads():
    g = np.load(os.path.join(GOLDEN, "encoder_tiny.npz"))
    nb = int(g["num_blocks"])
    sd = so.make_state_dict(so.SanmConfig(num_blocks=nb), seed=int(g["seed"]))
    m = CpuModel(num_blocks=nb, state_dict=sd, threads=4)
    out = m.encoder(g["feats"], g["lens"])
    assert np.max(np.abs(out - g["out"])) <= 2e-4
    for b, n in enumerate(g["lens"]):
        assert not out[b, n:].any()
    m.set_threads(1)
    out1 = m.encoder(g["feats"], g["lens"])
    assert np.array_equal(out1, out)  # every output element is summed by one thread in a fixed order
    m.close()


def test_encoder_against_the_torch_oracle_trained_like_norms():
    cfg = so.SanmConfig(num_blocks=3)
    sd = so.make_state_dict(cfg, seed=5)
    g = torch.Generator().manual_seed(6)
    for k in sd:
        if ".norm" in k or k.startswith("after_norm"):
            sd[k] = (sd[k] + 0.3 * torch.randn(sd[k].shape, generator=g)).float()
    feats = so.synth_feats(3, 37, 560, seed=7)
    lens = torch.tensor([37, 20, 0], dtype=torch.int32)
    want, _ = so.encoder_forward(feats.double(), lens, {k: v.double() for k, v in sd.items()}, cfg)
    m = CpuModel(num_blocks=3, state_dict={"encoder." + k: v for k, v in sd.items()})  # the prefix is accepted
    out = m.encoder(feats.numpy(), lens.numpy())
    assert np.max(np.abs(out - want.numpy())) <= 2e-4
    m.close()


def test_refusals():
    m = CpuModel(num_blocks=2)
    with pytest.raises(RuntimeError, match="not loaded"):
        m.encoder(np.zeros((1, 4, 560), dtype=np.float32), [4])
    sd = so.make_state_dict(so.SanmConfig(num_blocks=2), seed=1)
    del sd["encoders.0.feed_forward.w_2.bias"]
    with pytest.raises(RuntimeError, match="missing tensor encoders.0.feed_forward.w_2.bias"):
        CpuModel(num_blocks=2, state_dict=sd)
    m.close()
