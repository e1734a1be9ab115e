"""CPU tests of the oracle itself: golden fixtures, structural and analytic known answers.

The reference ships no golden vectors for this path (parity unpinned); the fixtures were produced by
tests/golden/make_golden.py, which also checks the fbank oracle against transformers' Kaldi-fbank
implementation at generation time.
"""
import glob
import math
import os

import numpy as np
import pytest
import torch

from oracle import fbank_oracle as fo
from oracle import sanm_oracle as so


def test_frame_counts():
    assert fo.num_frames(480000) == 2998
    assert fo.num_lfr_frames(2998) == 500
    assert fo.num_frames(48000) == 298 and fo.num_lfr_frames(298) == 50
    assert fo.num_frames(399) == 0 and fo.num_frames(400) == 1 and fo.num_frames(559) == 1 and fo.num_frames(560) == 2
    assert fo.num_lfr_frames(0) == 0 and fo.num_lfr_frames(1) == 1 and fo.num_lfr_frames(6) == 1 and fo.num_lfr_frames(7) == 2


def test_frame_indices_exact():
    x = np.arange(2000, dtype=np.float64)
    fr = fo.frame_signal(x)
    assert fr.shape == (11, 400)
    for f in range(11):
        assert fr[f, 0] == 160 * f and fr[f, -1] == 160 * f + 399


def test_silence_and_dc_hit_the_log_floor():
    floor = math.log(fo.LOG_FLOOR)
    z = fo.kaldi_fbank(np.zeros(16000, dtype=np.int16))
    assert z.shape == (98, 80) and np.all(z == floor)
    dc = fo.kaldi_fbank(np.full(16000, 1234, dtype=np.int16))
    assert np.allclose(dc, floor, atol=1e-9)


def test_tone_peaks_in_the_right_filter():
    t = np.arange(16000) / 16000.0
    x = np.rint(8000 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.int16)
    m = fo.kaldi_fbank(x)
    banks = fo.mel_banks()
    expect = int(np.argmax(banks[:, 32]))  # 1 kHz = bin 32
    assert np.all(np.argmax(m, axis=1) == expect)


def test_float_pcm_equals_int16_pcm():
    x = fo.synth_pcm("gauss", 4000, 3)
    a = fo.kaldi_fbank(x)
    b = fo.kaldi_fbank((x / 32768.0).astype(np.float32))
    assert np.array_equal(a, b)


def test_lfr_index_pattern():
    t, d = 14, 3
    feats = np.repeat(np.arange(t, dtype=np.float64)[:, None], d, axis=1)
    out = fo.apply_lfr(feats)
    assert out.shape == (3, 7 * d)
    rows = out.reshape(3, 7, d)[:, :, 0]
    assert rows[0].tolist() == [0, 0, 0, 0, 1, 2, 3]
    assert rows[1].tolist() == [3, 4, 5, 6, 7, 8, 9]
    assert rows[2].tolist() == [9, 10, 11, 12, 13, 13, 13]
    one = fo.apply_lfr(np.ones((1, 2)))
    assert one.shape == (1, 14) and np.all(one == 1)


def test_cmvn_is_shift_then_scale():
    rng = np.random.default_rng(0)
    x = rng.normal(size=(5, 560))
    mu, sd = x.mean(0), x.std(0)
    y = fo.apply_cmvn(x, -mu, 1.0 / sd)
    assert np.allclose(y.mean(0), 0, atol=1e-12) and np.allclose(y.std(0), 1, atol=1e-12)


def test_load_cmvn_roundtrip(tmp_path):
    shift = np.linspace(-9, -8, 560)
    scale = np.linspace(0.1, 0.2, 560)
    p = tmp_path / "am.mvn"
    p.write_text(
        "<Nnet>\n<Splice> 560 560\n[ 0 ]\n<AddShift> 560 560\n<LearnRateCoef> 0 [ "
        + " ".join(f"{v:.10g}" for v in shift) + " ]\n<Rescale> 560 560\n<LearnRateCoef> 0 [ "
        + " ".join(f"{v:.10g}" for v in scale) + " ]\n</Nnet>\n")
    s, c = fo.load_cmvn(str(p))
    assert np.allclose(s, shift) and np.allclose(c, scale)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "fbank_*.npz"))))
def test_fbank_golden(path):
    g = np.load(path)
    logmel = fo.kaldi_fbank(g["pcm"])
    assert logmel.shape == g["logmel"].shape
    assert np.max(np.abs(logmel - g["logmel"])) < 2e-5  # fixtures are stored in float32
    feats = fo.apply_cmvn(fo.apply_lfr(logmel), g["shift"].astype(np.float64), g["scale"].astype(np.float64))
    assert feats.shape == g["feats"].shape
    assert np.max(np.abs(feats - g["feats"])) < 2e-5


def test_frontend_batch_padding():
    a = fo.synth_pcm("gauss", 3000, 1)
    b = fo.synth_pcm("gauss", 1200, 2)
    batch = np.zeros((3, 3000), dtype=np.int16)
    batch[0, :3000] = a
    batch[1, :1200] = b
    feats, lens = fo.frontend(batch, [3000, 1200, 100])
    assert lens.tolist() == [3, 1, 0]
    assert feats.shape == (3, 3, 560)
    assert np.all(feats[1, 1:] == 0) and np.all(feats[2] == 0)


def test_encoder_tiny_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "encoder_tiny.npz"))
    cfg = so.SanmConfig(num_blocks=int(g["num_blocks"]))
    sd = so.make_state_dict(cfg, seed=int(g["seed"]), dtype=torch.float32)
    out, olens = so.encoder_forward(torch.from_numpy(g["feats"]), torch.from_numpy(g["lens"]), sd, cfg)
    assert olens.tolist() == g["olens"].tolist()
    assert np.max(np.abs(out.numpy() - g["out"])) < 2e-4  # fp32 run vs the fp64 fixture


def test_encoder_matches_module_style_implementation():
    """Independent restatement with torch.nn modules wired like FunASR's classes (Conv1d + ConstantPad1d)."""
    cfg = so.SanmConfig(num_blocks=2)
    sd = so.make_state_dict(cfg, seed=3, dtype=torch.float64)
    feats = so.synth_feats(2, 23, 560, seed=5, dtype=torch.float64)
    lens = torch.tensor([23, 9], dtype=torch.int32)
    want, _ = so.encoder_forward(feats, lens, sd, cfg)

    d, h = 512, 4
    mask = (torch.arange(23)[None, :] < lens[:, None]).double()
    x = feats * d ** 0.5 + so.sinusoidal_pe(23, 560, torch.float64)[None]
    for li, p in enumerate(so.layer_prefixes(cfg)):
        din = 560 if li == 0 else d
        n1 = torch.nn.LayerNorm(din, eps=1e-12).double()
        n1.weight.data, n1.bias.data = sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"]
        lin = torch.nn.Linear(din, 3 * d).double()
        lin.weight.data, lin.bias.data = sd[f"{p}.self_attn.linear_q_k_v.weight"], sd[f"{p}.self_attn.linear_q_k_v.bias"]
        conv = torch.nn.Conv1d(d, d, 11, groups=d, bias=False).double()
        conv.weight.data = sd[f"{p}.self_attn.fsmn_block.weight"]
        pad = torch.nn.ConstantPad1d((5, 5), 0.0)
        lo = torch.nn.Linear(d, d).double()
        lo.weight.data, lo.bias.data = sd[f"{p}.self_attn.linear_out.weight"], sd[f"{p}.self_attn.linear_out.bias"]
        with torch.no_grad():
            res = x
            y = n1(x)
            q, k, v = torch.split(lin(y), d, dim=-1)
            vm = v * mask[:, :, None]
            mem = (conv(pad(vm.transpose(1, 2))).transpose(1, 2) + vm) * mask[:, :, None]
            outs = []
            for b in range(2):
                heads = []
                for hh in range(h):
                    sl = slice(hh * 128, (hh + 1) * 128)
                    sc = (q[b, :, sl] * 128 ** -0.5) @ k[b, :, sl].T
                    sc[:, int(lens[b]):] = float("-inf")
                    heads.append(torch.softmax(sc, -1) @ v[b, :, sl])
                outs.append(torch.cat(heads, -1))
            att = lo(torch.stack(outs)) + mem
            x = res + att if din == d else att
            n2 = torch.nn.LayerNorm(d, eps=1e-12).double()
            n2.weight.data, n2.bias.data = sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"]
            hdn = torch.relu(torch.nn.functional.linear(n2(x), sd[f"{p}.feed_forward.w_1.weight"], sd[f"{p}.feed_forward.w_1.bias"]))
            x = x + torch.nn.functional.linear(hdn, sd[f"{p}.feed_forward.w_2.weight"], sd[f"{p}.feed_forward.w_2.bias"])
    with torch.no_grad():
        x = torch.nn.functional.layer_norm(x, (d,), sd["after_norm.weight"], sd["after_norm.bias"], 1e-12)
    x = x * mask[:, :, None]
    assert float((x - want).abs().max()) < 1e-9


def test_encoder_valid_rows_do_not_depend_on_padding():
    cfg = so.SanmConfig(num_blocks=2)
    sd = so.make_state_dict(cfg, seed=0)
    feats = so.synth_feats(1, 20, 560, seed=1)
    out_a, _ = so.encoder_forward(feats, torch.tensor([12]), sd, cfg)
    feats2 = feats.clone()
    feats2[:, 12:] = 123.0
    out_b, _ = so.encoder_forward(feats2, torch.tensor([12]), sd, cfg)
    assert torch.equal(out_a[:, :12], out_b[:, :12])
    assert float(out_a[:, 12:].abs().max()) == 0.0
