"""The oracles against independent implementations of the same published modules that ship in the image (torch's own
attention and convolution modules, transformers' HiFi-GAN port).  The reference holds no vectors for these paths, so a
second implementation that agrees to rounding is the strongest pin available: a slip in the restatement (a padding, a
slope, a mask) would show as a difference of the size of the signal, not of 1e-13."""
import importlib.util
import os

import pytest

_spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(mg)


def test_hifigan_oracle_equals_transformers_port():
    pytest.importorskip("transformers")
    assert mg.witness_hifigan() < 1e-12
    # the deployed stage layout (8, 8, 2, 2 with kernels 16, 16, 4, 4), narrow so that it runs in seconds
    assert mg.witness_hifigan(initial_channel=64, rates=(8, 8, 2, 2), kernels=(16, 16, 4, 4), frames=12, seed=19) < 1e-12


def test_sanm_attention_oracle_equals_torch_modules():
    err, fsmn_size = mg.witness_sanm_attention()
    assert err < 1e-12 and fsmn_size > 1e-3
    # first layer shape is the same module at d_model input; a ragged batch with a one-frame clip
    err, _ = mg.witness_sanm_attention(b=3, t=33, lens=(33, 1, 12), seed=23)
    assert err < 1e-12
