"""GPU parity tests: the HIP path, called through the C ABI (ctypes), against the CPU oracle and the
committed golden fixtures.  Tolerances follow the stated bars: frame indices / counts exact, log-mel and
LFR/CMVN features <= 1e-3 absolute (float32 kernels vs float64 oracle), encoder output <= 1e-2 absolute
(fp16 MFMA operands with fp32 accumulation vs float64/float32 oracle; bf16 operands are held to a looser bound).
"""
import glob
import math
import os

import numpy as np
import pytest
import torch

from oracle import fbank_oracle as fo
from oracle import sanm_oracle as so

pytestmark = pytest.mark.gpu

FBANK_TOL = 1e-3
ENC_TOL = 1e-2       # the stated bar; met with fp16 MFMA operands (the default compute dtype)
ENC_TOL_BF16 = 6e-2  # bf16 operands: rounding alone gives 1.7e-2 (2 blocks) .. 3.1e-2 (50 blocks) on these
#                      fixtures in a CPU emulation with exact accumulation, so this mode is below the bar
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def pkg():
    import toolbox_for_asr_and_tts_amd as p

    p.load()  # raises if the HIP library is missing: there is no fallback to test instead
    return p


@pytest.fixture(scope="module")
def fe(pkg):
    return pkg.WavFrontend()


def dev(x):
    return torch.as_tensor(x).to("cuda")


def batchify(clips):
    n = max(len(c) for c in clips)
    n = (n + 7) // 8 * 8
    arr = np.zeros((len(clips), max(n, 8)), dtype=clips[0].dtype)
    for i, c in enumerate(clips):
        arr[i, : len(c)] = c
    return arr, [len(c) for c in clips]


# ---- fbank -------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "fbank_*.npz"))))
def test_fbank_golden(fe, path):
    g = np.load(path)
    pcm = g["pcm"]
    out, nfr = fe.forward_fbank(dev(pcm[None]), [len(pcm)])
    assert int(nfr[0]) == g["logmel"].shape[0] == fo.num_frames(len(pcm))
    got = out[0].cpu().numpy()
    assert got.shape == g["logmel"].shape
    assert np.max(np.abs(got - g["logmel"])) <= FBANK_TOL


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "fbank_*.npz"))))
def test_frontend_golden_fused_and_split(fe, path):
    g = np.load(path)
    pcm = g["pcm"]
    fe.set_cmvn(g["shift"], g["scale"])
    try:
        feats, lens = fe.forward(dev(pcm[None]), [len(pcm)])
        assert int(lens[0]) == g["feats"].shape[0]
        got = feats[0].cpu().numpy()
        assert got.shape == g["feats"].shape
        assert np.max(np.abs(got - g["feats"])) <= FBANK_TOL
        # split path: fbank kernel then lfr_cmvn kernel must give the same numbers as the fused kernel
        logmel, nfr = fe.forward_fbank(dev(pcm[None]), [len(pcm)])
        feats2, lens2 = fe.forward_lfr_cmvn(logmel, nfr.cpu())
        assert torch.equal(lens2, lens)
        assert torch.equal(feats2, feats)
    finally:
        fe.set_cmvn(np.zeros(560, np.float32), np.ones(560, np.float32))


def test_fbank_ragged_batch_and_edges(fe):
    lens = [0, 1, 399, 400, 401, 559, 560, 1367, 8000, 16001, 3203, 720]
    kinds = ["gauss", "uniform", "bursts", "chirp"]
    clips = [fo.synth_pcm(kinds[i % 4], n, 40 + i) if n else np.zeros(0, np.int16) for i, n in enumerate(lens)]
    arr, ln = batchify(clips)
    assert ln == lens
    out, nfr = fe.forward_fbank(dev(arr), ln)
    want, wn = fo.fbank_batch(arr, ln)
    assert nfr.cpu().tolist() == wn.tolist() == [fo.num_frames(n) for n in lens]
    got = out.cpu().numpy()
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= FBANK_TOL
    for b, t in enumerate(wn):
        assert np.all(got[b, t:] == 0.0), "rows past n_frames must be zero"
    # fused frontend on the same ragged batch
    shift = np.linspace(-12, -6, 560).astype(np.float32)
    scale = np.linspace(0.2, 0.5, 560).astype(np.float32)
    fe.set_cmvn(shift, scale)
    try:
        feats, flens = fe.forward(dev(arr), ln)
        wf, wl = fo.frontend(arr, ln, shift, scale)
        assert flens.cpu().tolist() == wl.tolist()
        gotf = feats.cpu().numpy()
        assert gotf.shape == wf.shape
        assert np.max(np.abs(gotf - wf)) <= FBANK_TOL
        for b, t in enumerate(wl):
            assert np.all(gotf[b, t:] == 0.0)
    finally:
        fe.set_cmvn(np.zeros(560, np.float32), np.ones(560, np.float32))


def test_fbank_float_input_equals_int16_input(fe):
    pcm = fo.synth_pcm("gauss", 16000, 7)
    a, _ = fe.forward_fbank(dev(pcm[None]), [16000])
    b, _ = fe.forward_fbank(dev((pcm / 32768.0).astype(np.float32)[None]), [16000])
    assert torch.equal(a, b)


def test_fbank_unaligned_rows_take_the_scalar_path(fe):
    pcm = fo.synth_pcm("gauss", 4001, 8)
    big = np.zeros((2, 4003), dtype=np.int16)  # odd row stride -> rows are not 16-byte aligned
    big[0, :4001] = pcm
    big[1, :3000] = pcm[:3000]
    out, nfr = fe.forward_fbank(dev(big), [4001, 3000])
    want, wn = fo.fbank_batch(big, [4001, 3000])
    assert nfr.cpu().tolist() == wn.tolist()
    assert np.max(np.abs(out.cpu().numpy() - want)) <= FBANK_TOL


def test_fbank_analytic_known_answers(fe):
    floor = math.log(fo.LOG_FLOOR)
    z = np.zeros((1, 16000), np.int16)
    out, _ = fe.forward_fbank(dev(z), [16000])
    assert torch.all(out == out[0, 0, 0]) and abs(float(out[0, 0, 0]) - floor) < 1e-5
    dc = np.full((1, 16000), 1234, np.int16)
    out_dc, _ = fe.forward_fbank(dev(dc), [16000])
    assert torch.equal(out_dc, out), "a constant signal must vanish after DC removal"
    t = np.arange(16000) / 16000.0
    tone = np.rint(8000 * np.sin(2 * np.pi * 1000.0 * t)).astype(np.int16)[None]
    out, _ = fe.forward_fbank(dev(tone), [16000])
    expect = int(np.argmax(fo.mel_banks()[:, 32]))
    assert torch.all(out[0].argmax(dim=1) == expect)


def test_fbank_full_size_properties(fe):
    """30 s clips (T = 2998, T' = 500): size-independent properties instead of a slow oracle sweep."""
    n = 480000
    b = 6
    arr = np.stack([fo.synth_pcm("bench", n, 1234 + i) for i in range(b)])
    x = dev(arr)
    out, nfr = fe.forward_fbank(x, [n] * b)
    assert out.shape == (b, 2998, 80) and nfr.cpu().tolist() == [2998] * b
    assert bool(torch.isfinite(out).all())
    # (1) the zero prefix (first 10 %) sits exactly on the log floor
    zp = out[:, : (n // 10 - 400) // 160]
    assert torch.all(zp == zp[0, 0, 0]) and abs(float(zp[0, 0, 0]) - math.log(fo.LOG_FLOOR)) < 1e-5
    # (2) shifting the signal by one hop shifts the frames by one, bit for bit
    shifted = torch.zeros_like(x)
    shifted[:, : n - 160] = x[:, 160:]
    out_s, _ = fe.forward_fbank(shifted, [n - 160] * b)
    assert torch.equal(out_s[:, :2997], out[:, 1:2998])
    # (3) results do not depend on the batch position or on the neighbours
    out_r, _ = fe.forward_fbank(x.flip(0).contiguous(), [n] * b)
    assert torch.equal(out_r.flip(0), out)
    # (4) spot rows against the oracle
    want = fo.kaldi_fbank(arr[2])
    got = out[2].cpu().numpy()
    assert np.max(np.abs(got - want)) <= FBANK_TOL
    # (5) fused LFR/CMVN output is a pure gather of the fbank output (identity CMVN): exact
    feats, fl = fe.forward(x, [n] * b)
    assert feats.shape == (b, 500, 560) and fl.cpu().tolist() == [500] * b
    idx = np.clip(6 * np.arange(500)[:, None] + np.arange(7)[None, :] - 3, 0, 2997)
    gathered = out[:, torch.as_tensor(idx, device="cuda")].reshape(b, 500, 560)
    assert torch.equal(feats, gathered)


@pytest.mark.parametrize("n_mels,lfr_m,lfr_n", [(40, 5, 3), (64, 1, 1), (32, 7, 6), (80, 5, 1)])  # the last: fsmn-vad's frontend
def test_fbank_other_mel_layouts(pkg, n_mels, lfr_m, lfr_n):
    """Filter counts other than 80 go through the kernel's run-time mel loops (the 80-filter layout has its trip
    counts compiled in) and other LFR shapes through the scatter: log-mel and features against the oracle."""
    fe2 = pkg.WavFrontend(n_mels=n_mels, lfr_m=lfr_m, lfr_n=lfr_n)
    lens = [16000, 5203, 400, 399, 24000]
    pcm = np.zeros((len(lens), max(lens)), np.int16)
    for i, n in enumerate(lens):
        pcm[i, :n] = fo.synth_pcm(["gauss", "chirp", "bursts", "uniform", "bench"][i], n, 40 + i)
    banks = fo.mel_banks(n_mels=n_mels)
    got, nfr = fe2.forward_fbank(dev(pcm), lens)
    feats, flens = fe2.forward(dev(pcm), lens)
    torch.cuda.synchronize()
    for i, n in enumerate(lens):
        want = fo.kaldi_fbank(pcm[i, :n], banks=banks) if n >= 400 else np.zeros((0, n_mels))
        t = want.shape[0]
        assert int(nfr[i]) == t
        if t == 0:
            assert int(flens[i]) == 0
            continue
        assert np.abs(got[i, :t].cpu().numpy() - want).max() <= FBANK_TOL
        wl = fo.apply_lfr(want, lfr_m, lfr_n)
        assert int(flens[i]) == wl.shape[0]
        assert np.abs(feats[i, :wl.shape[0]].cpu().numpy() - wl).max() <= FBANK_TOL


def test_too_wide_mel_filters_are_refused(pkg):
    """23 filters over 20 Hz .. 8 kHz are wider than the kernel's weight table: creation fails with a message."""
    with pytest.raises(Exception, match="mel filters too wide"):
        pkg.WavFrontend(n_mels=23)


def test_frontend_bf16_output(fe):
    pcm = fo.synth_pcm("gauss", 16000, 9)
    f32, _ = fe.forward(dev(pcm[None]), [16000])
    bf, _ = fe.forward(dev(pcm[None]), [16000], out_dtype=torch.bfloat16)
    assert torch.equal(bf, f32.to(torch.bfloat16))


def test_bad_arguments_raise(pkg, fe):
    with pytest.raises(ValueError):
        fe.forward_fbank(torch.zeros(2, 800, dtype=torch.int16), [800, 800])  # host tensor
    with pytest.raises(ValueError):
        fe.forward_fbank(dev(np.zeros((2, 800), np.int16)), [800, 801])  # length past the row
    with pytest.raises(ValueError):
        fe.forward_fbank(dev(np.zeros((2, 800), np.float64)), [800, 800])
    enc = pkg.SANMEncoder(num_blocks=1)
    with pytest.raises(pkg.AsrfeError):  # weights not loaded
        enc.forward(torch.zeros(1, 8, 560, device="cuda"), [8])


# ---- encoder ---------------------------------------------------------------------------------------
# Calls of up to enc_small_batch_rows rows (2048 by default) take the latency launch plan, larger ones the throughput
# plan; the oracle tests below run the sizes they were written for under BOTH (0 switches the latency plan off).
PLANS = [pytest.param(0, id="throughput-plan"), pytest.param(1024, id="latency-plan")]


def run_encoder(pkg, num_blocks, seed, feats, lens, out_dtype=torch.float32, compute_dtype="float16", small_rows=None):
    cfg = so.SanmConfig(num_blocks=num_blocks)
    sd = so.make_state_dict(cfg, seed=seed, dtype=torch.float32)
    enc = pkg.SANMEncoder(num_blocks=num_blocks, compute_dtype=compute_dtype, small_batch_rows=small_rows)
    enc.load_state_dict(sd)
    out, olens, _ = enc.forward(feats.to("cuda"), lens, out_dtype=out_dtype)
    torch.cuda.synchronize()
    return out, olens, sd, cfg


@pytest.mark.parametrize("small_rows", PLANS)
def test_encoder_tiny_golden(pkg, small_rows):
    g = np.load(os.path.join(GOLDEN, "encoder_tiny.npz"))
    feats = torch.from_numpy(g["feats"])
    out, olens, _, _ = run_encoder(pkg, int(g["num_blocks"]), int(g["seed"]), feats, torch.from_numpy(g["lens"]),
                                   small_rows=small_rows)
    assert olens.cpu().tolist() == g["olens"].tolist()
    got = out.cpu().numpy()
    err = np.max(np.abs(got - g["out"]))
    assert err <= ENC_TOL, err
    assert np.all(got[1, int(g["lens"][1]):] == 0.0)


def test_encoder_bf16_mode(pkg):
    """bf16 MFMA operands are selectable; their error is rounding-limited and checked against a looser bound."""
    g = np.load(os.path.join(GOLDEN, "encoder_tiny.npz"))
    feats = torch.from_numpy(g["feats"])
    out, _, _, _ = run_encoder(pkg, int(g["num_blocks"]), int(g["seed"]), feats, torch.from_numpy(g["lens"]),
                               compute_dtype="bfloat16")
    err = np.max(np.abs(out.cpu().numpy() - g["out"]))
    assert err <= ENC_TOL_BF16, err
    g = np.load(os.path.join(GOLDEN, "encoder_probe_50l.npz"))
    feats = so.synth_feats(2, 96, 560, seed=int(g["feats_seed"]))
    out, _, _, _ = run_encoder(pkg, 50, int(g["seed"]), feats, torch.from_numpy(g["lens"]), compute_dtype="bfloat16")
    got = out.cpu().numpy()
    e = max(np.max(np.abs(got[0, [0, 1, 47, 95]] - g["rows_b0"])), np.max(np.abs(got[1, [0, 30, 60]] - g["rows_b1"])))
    assert e <= ENC_TOL_BF16, e


@pytest.mark.parametrize("small_rows", PLANS)
def test_encoder_one_layer_vs_oracle_various_lengths(pkg, small_rows):
    """Exercises the key mask, the FSMN edge handling and row tails: T' not a multiple of 64 / 128."""
    for tp, lens in [(1, [1]), (7, [7, 3]), (65, [65, 64, 1]), (130, [130, 77]), (200, [129, 200, 5, 64])]:
        feats = so.synth_feats(len(lens), tp, 560, seed=tp)
        lt = torch.tensor(lens, dtype=torch.int32)
        out, olens, sd, cfg = run_encoder(pkg, 1, 5, feats, lt, small_rows=small_rows)
        want, wl = so.encoder_forward(feats.double(), lt, {k: v.double() for k, v in sd.items()}, cfg)
        assert olens.cpu().tolist() == wl.tolist()
        err = float((out.cpu().double() - want).abs().max())
        assert err <= ENC_TOL, (tp, lens, err)


@pytest.mark.parametrize("small_rows", PLANS)
def test_encoder_random_shapes_vs_oracle(pkg, small_rows):
    """Seeded random batches (1-7 utterances, T' up to 330, lengths from 1 to T'), three layers so that the layers
    hand the residual stream on in the kernels' own layout: row counts that end anywhere inside a 32-row group, a
    128-row tile or a key tile, utterances of one frame next to full ones."""
    rng = np.random.default_rng(20240607)
    cfg = so.SanmConfig(num_blocks=3)
    sd = so.make_state_dict(cfg, seed=8, dtype=torch.float32)
    enc = pkg.SANMEncoder(num_blocks=3, small_batch_rows=small_rows)
    enc.load_state_dict(sd)
    sdd = {k: v.double() for k, v in sd.items()}
    for case in range(10):
        b = int(rng.integers(1, 8))
        tp = int(rng.integers(1, 331))
        lens = [int(v) for v in rng.integers(1, tp + 1, size=b)]
        lens[int(rng.integers(0, b))] = tp
        feats = so.synth_feats(b, tp, 560, seed=100 + case)
        lt = torch.tensor(lens, dtype=torch.int32)
        out, olens, _ = enc.forward(feats.cuda(), lt)
        torch.cuda.synchronize()
        want, wl = so.encoder_forward(feats.double(), lt, sdd, cfg)
        assert olens.cpu().tolist() == wl.tolist()
        err = float((out.cpu().double() - want).abs().max())
        assert err <= ENC_TOL, (case, b, tp, lens, err)
        for i, n in enumerate(lens):
            assert float(out[i, n:].abs().max()) == 0.0 if n < tp else True


@pytest.mark.parametrize("small_rows", PLANS)
def test_encoder_50_layers_probe(pkg, small_rows):
    g = np.load(os.path.join(GOLDEN, "encoder_probe_50l.npz"))
    feats = so.synth_feats(2, 96, 560, seed=int(g["feats_seed"]))
    out, olens, _, _ = run_encoder(pkg, 50, int(g["seed"]), feats, torch.from_numpy(g["lens"]), small_rows=small_rows)
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    e0 = np.max(np.abs(got[0, [0, 1, 47, 95]] - g["rows_b0"]))
    e1 = np.max(np.abs(got[1, [0, 30, 60]] - g["rows_b1"]))
    assert max(e0, e1) <= ENC_TOL, (e0, e1)
    assert np.max(np.abs(got.sum(axis=(0, 1)) - g["col_sums"])) <= ENC_TOL * 157  # 157 valid rows
    assert abs(np.abs(got).mean() - float(g["abs_mean"])) <= 1e-3


def test_encoder_full_length_properties(pkg):
    """T' = 500 (30 s), 3 layers: padding independence, batch independence, determinism, bf16 output."""
    tp = 500
    feats = so.synth_feats(3, tp, 560, seed=21)
    lens = torch.tensor([500, 333, 64], dtype=torch.int32)
    out, _, sd, cfg = run_encoder(pkg, 3, 1, feats, lens)
    want, _ = so.encoder_forward(feats, lens, sd, cfg)
    assert float((out.cpu() - want).abs().max()) <= ENC_TOL
    enc = pkg.SANMEncoder(num_blocks=3)
    enc.load_state_dict(sd)
    f2 = feats.clone()
    f2[1, 333:] = 50.0  # garbage in the padding must not reach valid rows
    out2, _, _ = enc.forward(f2.cuda(), lens)
    assert torch.equal(out2, out)
    out3, _, _ = enc.forward(feats.flip(0).contiguous().cuda(), lens.flip(0))
    assert torch.equal(out3.flip(0), out)
    outb, _, _ = enc.forward(feats.cuda(), lens, out_dtype=torch.bfloat16)
    assert torch.equal(outb, out.to(torch.bfloat16))
    outh, _, _ = enc.forward(feats.cuda(), lens, out_dtype=torch.float16)
    assert torch.equal(outh, out.to(torch.float16))
    assert float(out[1, 333:].abs().max()) == 0.0 and float(out[2, 64:].abs().max()) == 0.0


def test_engine_end_to_end_matches_oracle(pkg):
    """PCM -> frontend -> 2-layer encoder through AsrEngine (preallocated buffers) vs the oracle chain."""
    n = 48000  # 3 s: T = 298, T' = 50
    clips = np.stack([fo.synth_pcm("bursts", n, 70), fo.synth_pcm("gauss", n, 71), fo.synth_pcm("chirp", n, 72)])
    lens = [48000, 30000, 16000]
    rng = np.random.default_rng(5)
    shift = -rng.uniform(6, 14, 560)
    scale = 1.0 / rng.uniform(1.5, 4.5, 560)
    cfg_o = so.SanmConfig(num_blocks=2)
    sd = so.make_state_dict(cfg_o, seed=2)
    cfg = pkg.default_config()
    cfg.enc_num_blocks = 2
    eng = pkg.AsrEngine(batch=3, max_samples=n, cfg=cfg, cmvn=(shift.astype(np.float32), scale.astype(np.float32)),
                        state_dict=sd)
    out, olens = eng.run(dev(clips), dev(np.array(lens, np.int32)))
    torch.cuda.synchronize()
    wf, wl = fo.frontend(clips, lens, shift, scale)
    assert olens.cpu().tolist() == wl.tolist()
    want, _ = so.encoder_forward(torch.from_numpy(wf).float(), torch.from_numpy(wl), sd, cfg_o)
    assert float((out.cpu() - want).abs().max()) <= ENC_TOL


def _trained_like(sd, seed):
    """LayerNorm scale / shift and all biases away from their init values, as in a trained checkpoint: the
    library folds norm1 / norm2 into the linear layers behind them, which the gamma = 1, beta = 0 goldens cannot see."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        if "norm" in k and k.endswith(".weight"):
            out[k] = v * (1.0 + 0.3 * torch.randn(v.shape, generator=g, dtype=v.dtype))
        elif "norm" in k and k.endswith(".bias"):
            out[k] = v + 0.2 * torch.randn(v.shape, generator=g, dtype=v.dtype)
        else:
            out[k] = v
    return out


@pytest.mark.parametrize("small_rows", PLANS)
def test_encoder_nontrivial_layernorm_parameters(pkg, small_rows):
    """3 layers with random LayerNorm gamma / beta (including after_norm) against the oracle."""
    cfg = so.SanmConfig(num_blocks=3)
    sd = _trained_like(so.make_state_dict(cfg, seed=9, dtype=torch.float32), seed=10)
    lens = torch.tensor([150, 97, 128], dtype=torch.int32)
    feats = so.synth_feats(3, 150, 560, seed=33)
    enc = pkg.SANMEncoder(num_blocks=3, small_batch_rows=small_rows)
    enc.load_state_dict(sd)
    out, olens, _ = enc.forward(feats.cuda(), lens)
    want, wl = so.encoder_forward(feats.double(), lens, {k: v.double() for k, v in sd.items()}, cfg)
    assert olens.cpu().tolist() == wl.tolist()
    err = float((out.cpu().double() - want).abs().max())
    assert err <= ENC_TOL, err


@pytest.mark.parametrize("small_rows", PLANS)
@pytest.mark.parametrize("ff", [256, 1280, 3072, 5888])
def test_encoder_other_feed_forward_widths(pkg, ff, small_rows):
    """linear_units other than 2048: the weight stream of the post-attention kernel has ff / 64 hidden chunks, and
    above 5632 units its LDS has no room for the fused QKV projection, so every layer launches its own (5888 takes
    that plan).  3 layers against the oracle."""
    cfg = so.SanmConfig(num_blocks=3, linear_units=ff)
    sd = _trained_like(so.make_state_dict(cfg, seed=20 + ff, dtype=torch.float32), seed=21)
    lens = torch.tensor([70, 33, 129], dtype=torch.int32)
    feats = so.synth_feats(3, 129, 560, seed=5)
    # (the latency plan takes linear_units in steps of 512 up to 3072: of these widths only 3072; the others fall back)
    enc = pkg.SANMEncoder(num_blocks=3, linear_units=ff, small_batch_rows=small_rows)
    enc.load_state_dict(sd)
    out, olens, _ = enc.forward(feats.cuda(), lens)
    want, wl = so.encoder_forward(feats.double(), lens, {k: v.double() for k, v in sd.items()}, cfg)
    assert olens.cpu().tolist() == wl.tolist()
    err = float((out.cpu().double() - want).abs().max())
    assert err <= ENC_TOL, (ff, err)


def test_post_attention_kernel_vs_fp64(pkg, monkeypatch):
    """The fused kernel behind attention (output projection + FSMN + residual, norm2, feed-forward, next-layer
    normalisation or next-layer QKV projection) on its own, through the debug hook: x_out within 2e-3, the normalised
    y within 1e-2 and the next layer's q, k, v within 2e-2 of an fp64 evaluation on the same 16-bit-rounded inputs;
    row counts around the 32 / 128 row tile edges."""
    import ctypes as C
    # (the 32x32x16 form of the kernel, which serves every launch plan; a context's weight stream is packed for one form)
    monkeypatch.setenv("ASRFE_MFMA16", "0")
    cfg = so.SanmConfig(num_blocks=2)
    sd = _trained_like(so.make_state_dict(cfg, seed=4, dtype=torch.float32), seed=6)
    enc = pkg.SANMEncoder(num_blocks=2)
    enc.load_state_dict(sd)
    ctx = enc.ctx
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    w = {k: v.double().cuda() for k, v in sd.items()}
    p = "encoders0.0."
    for m, with_x in [(1, True), (31, False), (32, True), (129, True), (257, False), (1000, True)]:
        gen = torch.Generator().manual_seed(m)
        att = (torch.randn(m, 512, generator=gen) * 0.5).half().cuda()
        fsmn = (torch.randn(m, 512, generator=gen) * 0.5).half().cuda()
        x = (torch.randn(m, 512, generator=gen) * 2.0 + 0.3).cuda()
        xo = torch.empty_like(x)
        y = torch.empty(m, 512, dtype=torch.float16, device="cuda")
        ctx.check(ctx.lib.asrfe_debug_post_attn(ctx.handle, 0, att.data_ptr(), fsmn.data_ptr(),
                                                x.data_ptr() if with_x else None, xo.data_ptr(), y.data_ptr(), None, m, 0,
                                                stream))
        # the variant the encoder runs between layers: same x_out, and the next layer's q, k, v instead of y
        xo2 = torch.empty_like(x)
        qkv = torch.empty(m, 1536, dtype=torch.float16, device="cuda")
        ctx.check(ctx.lib.asrfe_debug_post_attn(ctx.handle, 0, att.data_ptr(), fsmn.data_ptr(),
                                                x.data_ptr() if with_x else None, xo2.data_ptr(), None, qkv.data_ptr(), m, 0,
                                                stream))
        xm = att.double() @ w[p + "self_attn.linear_out.weight"].T + w[p + "self_attn.linear_out.bias"] + fsmn.double()
        if with_x:
            xm = xm + x.double()
        hh = torch.nn.functional.layer_norm(xm, (512,), w[p + "norm2.weight"], w[p + "norm2.bias"], 1e-12)
        hh = torch.relu(hh @ w[p + "feed_forward.w_1.weight"].T + w[p + "feed_forward.w_1.bias"])
        ref = xm + hh @ w[p + "feed_forward.w_2.weight"].T + w[p + "feed_forward.w_2.bias"]
        yref = torch.nn.functional.layer_norm(ref, (512,), None, None, 1e-12)
        assert float((xo.double() - ref).abs().max()) <= 2e-3, (m, with_x)
        assert float((y.double() - yref).abs().max()) <= 1e-2, (m, with_x)
        n1 = torch.nn.functional.layer_norm(ref, (512,), w["encoders.0.norm1.weight"], w["encoders.0.norm1.bias"], 1e-12)
        qref = n1 @ w["encoders.0.self_attn.linear_q_k_v.weight"].T + w["encoders.0.self_attn.linear_q_k_v.bias"]
        assert torch.equal(xo, xo2), (m, with_x)
        assert float((qkv.double() - qref).abs().max()) <= 2e-2, (m, with_x)
        # the residual stream in the kernel's register order (flags bit 9; what the encoder's throughput plan passes
        # from layer to layer), whole 128-row tiles per buffer.  The same sums in another order (the residual joins
        # the first 128 columns after the projection, the others before it): equal to fp32 rounding where no 16-bit
        # rounding of a normalised value flips, within 1e-3 where some do (each variant meets the fp64 bar by itself)
        mp = (m + 127) // 128 * 128
        xt = torch.zeros(mp, 512, device="cuda")
        xt[:m] = x
        xt = xt.view(mp // 32, 32, 16, 4, 2, 4).permute(0, 2, 3, 4, 1, 5).contiguous()
        xo3 = torch.empty_like(x)  # tiled in, row-major out: the encoder's last layer
        ctx.check(ctx.lib.asrfe_debug_post_attn(ctx.handle, 0, att.data_ptr(), fsmn.data_ptr(),
                                                xt.data_ptr() if with_x else None, xo3.data_ptr(), None, None, m, 512, stream))
        assert float((xo - xo3).abs().max()) <= 1e-3, (m, with_x)  # (16-bit roundings of n(x_mid) may flip)
        assert float((xo3.double() - ref).abs().max()) <= 2e-3, (m, with_x)
        xo4 = torch.full((mp // 32, 16, 4, 2, 32, 4), float("nan"), device="cuda")  # tiled in and out, with q, k, v
        qkv4 = torch.empty_like(qkv)
        ctx.check(ctx.lib.asrfe_debug_post_attn(ctx.handle, 0, att.data_ptr(), fsmn.data_ptr(),
                                                xt.data_ptr() if with_x else None, xo4.data_ptr(), None, qkv4.data_ptr(), m, 512,
                                                stream))
        assert torch.equal(xo3, xo4.permute(0, 4, 1, 2, 3, 5).reshape(mp, 512)[:m]), (m, with_x)
        assert float((qkv.float() - qkv4.float()).abs().max()) <= 4e-3, (m, with_x)
        assert float((qkv4.double() - qref).abs().max()) <= 2e-2, (m, with_x)


@pytest.mark.parametrize("dtype", ["float16", "bfloat16"])
def test_post_attention_kernel_16x16x32_vs_fp64(pkg, monkeypatch, dtype):
    """The same kernel on v_mfma_f32_16x16x32 (the throughput plan's default): tiled residual stream in ITS register
    order (per 32-row group and 32-column block four runs of 1 KiB: run 2 a + mh = columns 16 a + 4 q .. + 3 of rows
    16 mh + c for lane 16 q + c), x_out alone (row-major out: the last layer) and x_out tiled with the next layer's
    q, k, v; against fp64 on the same 16-bit-rounded inputs, and against the 32x32x16 form on the same inputs."""
    import ctypes as C
    cfg = so.SanmConfig(num_blocks=2)
    sd = _trained_like(so.make_state_dict(cfg, seed=4, dtype=torch.float32), seed=6)
    monkeypatch.setenv("ASRFE_MFMA16", "0")
    old = pkg.SANMEncoder(num_blocks=2, compute_dtype=dtype)
    old.load_state_dict(sd)
    monkeypatch.setenv("ASRFE_MFMA16", "1")
    enc = pkg.SANMEncoder(num_blocks=2, compute_dtype=dtype)
    enc.load_state_dict(sd)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    w = {k: v.double().cuda() for k, v in sd.items()}
    p = "encoders0.0."
    t16 = torch.float16 if dtype == "float16" else torch.bfloat16
    tol_x, tol_q = (2e-3, 2e-2) if dtype == "float16" else (1.6e-2, 1.6e-1)
    for m, with_x in [(1, True), (31, False), (32, True), (129, True), (257, False), (1000, True)]:
        gen = torch.Generator().manual_seed(m)
        att = (torch.randn(m, 512, generator=gen) * 0.5).to(t16).cuda()
        fsmn = (torch.randn(m, 512, generator=gen) * 0.5).to(t16).cuda()
        x = (torch.randn(m, 512, generator=gen) * 2.0 + 0.3).cuda()
        xm = att.double() @ w[p + "self_attn.linear_out.weight"].T + w[p + "self_attn.linear_out.bias"] + fsmn.double()
        if with_x:
            xm = xm + x.double()
        hh = torch.nn.functional.layer_norm(xm, (512,), w[p + "norm2.weight"], w[p + "norm2.bias"], 1e-12)
        hh = torch.relu(hh @ w[p + "feed_forward.w_1.weight"].T + w[p + "feed_forward.w_1.bias"])
        ref = xm + hh @ w[p + "feed_forward.w_2.weight"].T + w[p + "feed_forward.w_2.bias"]
        n1 = torch.nn.functional.layer_norm(ref, (512,), w["encoders.0.norm1.weight"], w["encoders.0.norm1.bias"], 1e-12)
        qref = n1 @ w["encoders.0.self_attn.linear_q_k_v.weight"].T + w["encoders.0.self_attn.linear_q_k_v.bias"]
        mp = (m + 127) // 128 * 128
        xp = torch.zeros(mp, 512, device="cuda")
        xp[:m] = x
        # row-major [group][mh][c][nt][a][q][e] -> [group][nt][a][mh][q][c][e]
        xt = xp.view(mp // 32, 2, 16, 16, 2, 4, 4).permute(0, 3, 4, 1, 5, 2, 6).contiguous()
        xo = torch.empty_like(x)
        enc.ctx.check(enc.ctx.lib.asrfe_debug_post_attn(enc.ctx.handle, 0, att.data_ptr(), fsmn.data_ptr(),
                                                        xt.data_ptr() if with_x else None, xo.data_ptr(), None, None, m, 512, stream))
        assert float((xo.double() - ref).abs().max()) <= tol_x, (m, with_x)
        xo_t = torch.full((mp // 32, 16, 2, 2, 4, 16, 4), float("nan"), device="cuda")
        qkv = torch.empty(m, 1536, dtype=t16, device="cuda")
        enc.ctx.check(enc.ctx.lib.asrfe_debug_post_attn(enc.ctx.handle, 0, att.data_ptr(), fsmn.data_ptr(),
                                                        xt.data_ptr() if with_x else None, xo_t.data_ptr(), None, qkv.data_ptr(), m, 512,
                                                        stream))
        assert torch.equal(xo, xo_t.permute(0, 3, 5, 1, 2, 4, 6).reshape(mp, 512)[:m]), (m, with_x)
        assert float((qkv.double() - qref).abs().max()) <= tol_q, (m, with_x)
        # the other form on the same inputs (its own tiled order): the same sums in another order
        xt_old = xp.view(mp // 32, 32, 16, 4, 2, 4).permute(0, 2, 3, 4, 1, 5).contiguous()
        xo_old = torch.empty_like(x)
        old.ctx.check(old.ctx.lib.asrfe_debug_post_attn(old.ctx.handle, 0, att.data_ptr(), fsmn.data_ptr(),
                                                        xt_old.data_ptr() if with_x else None, xo_old.data_ptr(), None, None, m, 512,
                                                        stream))
        assert float((xo - xo_old).abs().max()) <= (1e-3 if dtype == "float16" else 1.6e-2), (m, with_x)
    # what this form does not serve is refused, not computed wrongly
    y = torch.empty(8, 512, dtype=t16, device="cuda")
    rc = enc.ctx.lib.asrfe_debug_post_attn(enc.ctx.handle, 0, att.data_ptr(), fsmn.data_ptr(), None, xo.data_ptr(), y.data_ptr(),
                                           None, 8, 0, stream)
    assert rc == -1
    with pytest.raises(pkg.AsrfeError, match="tiled residual stream"):
        enc.ctx.check(rc)


def test_fused_and_unfused_encoder_paths_agree(pkg, monkeypatch):
    """Launch plans for the same layers: the throughput plan (the post-attention kernel also projects the next
    layer's q, k, v), ASRFE_SEPARATE_QKV=1 (it writes the normalised rows and every layer launches its own projection),
    ASRFE_UNFUSED_FFN=1 (separate LayerNorm / GEMM launches) and the latency plan (32 x 32 output tiles, the default
    at this size).  Each must meet the parity bar; they round to 16 bits at different points, so among themselves they
    agree to 4e-3, not bit for bit."""
    cfg = so.SanmConfig(num_blocks=3)
    sd = _trained_like(so.make_state_dict(cfg, seed=12, dtype=torch.float32), seed=13)
    lens = torch.tensor([130, 77], dtype=torch.int32)
    feats = so.synth_feats(2, 130, 560, seed=8)
    outs = []
    for unfused, separate, small, rowx in (("0", "0", 0, "0"), ("0", "1", 0, "0"), ("1", "0", 0, "0"), ("0", "0", 1024, "0"),
                                           ("0", "0", 0, "1")):
        monkeypatch.setenv("ASRFE_UNFUSED_FFN", unfused)
        monkeypatch.setenv("ASRFE_SEPARATE_QKV", separate)
        monkeypatch.setenv("ASRFE_ROW_MAJOR_X", rowx)
        enc = pkg.SANMEncoder(num_blocks=3, small_batch_rows=small)
        enc.load_state_dict(sd)
        out, _, _ = enc.forward(feats.cuda(), lens)
        outs.append(out.cpu())
    want, _ = so.encoder_forward(feats, lens, sd, cfg)
    for o in outs:
        assert float((o - want).abs().max()) <= ENC_TOL
    assert float((outs[0] - outs[1]).abs().max()) <= 4e-3
    assert float((outs[0] - outs[2]).abs().max()) <= 4e-3
    assert float((outs[0] - outs[3]).abs().max()) <= 4e-3
    assert not torch.equal(outs[0], outs[3])  # the latency plan really is another sequence of kernels
    # ASRFE_ROW_MAJOR_X=1 keeps the residual stream row-major between the post-attention launches (the throughput plan
    # otherwise hands it on in the kernel's register order and adds it at another point of the same sum); that plan also
    # takes the 32x32x16 form of the kernel, whose products accumulate in another order than the default 16x16x32 form
    assert float((outs[0] - outs[4]).abs().max()) <= 4e-3
    assert not torch.equal(outs[0], outs[4])


def test_engine_streams_give_identical_results(pkg):
    """Two chunks on two HIP streams, and the single-stream run of the same engine, are bit-identical."""
    n = 16000 * 4
    clips = np.stack([fo.synth_pcm(k, n, 80 + i) for i, k in enumerate(["bursts", "gauss", "chirp", "uniform", "bench"])])
    lens = dev(np.array([64000, 50000, 16000, 700, 33333], np.int32))
    sd = so.make_state_dict(so.SanmConfig(num_blocks=2), seed=3)
    cfg = pkg.default_config()
    cfg.enc_num_blocks = 2
    e2 = pkg.AsrEngine(batch=5, max_samples=n, cfg=cfg, state_dict=sd, n_streams=2)
    out2 = e2.run(dev(clips), lens)[0].clone()
    out1 = e2.run_single_stream(dev(clips), lens)[0].clone()
    torch.cuda.synchronize()
    assert torch.equal(out1, out2)
    e1 = pkg.AsrEngine(batch=5, max_samples=n, cfg=cfg, state_dict=sd, n_streams=1)
    out0 = e1.run(dev(clips), lens)[0]
    torch.cuda.synchronize()
    assert torch.equal(out0, out2)


def test_engine_staged_host_batches_match_direct_runs(pkg):
    """Batches staged from pinned host memory one ahead of the step (two device slots, a copy stream) give exactly
    what run() gives on device-resident copies of the same batches."""
    n = 16000 * 2
    sd = so.make_state_dict(so.SanmConfig(num_blocks=1), seed=3)
    cfg = pkg.default_config()
    cfg.enc_num_blocks = 1
    eng = pkg.AsrEngine(batch=3, max_samples=n, cfg=cfg, state_dict=sd, n_streams=2)
    batches = []
    for b in range(3):
        clips = np.stack([fo.synth_pcm(k, n, 10 * b + i) for i, k in enumerate(["gauss", "chirp", "bursts"])])
        lens = np.array([n, n - 1234 * (b + 1), 5000 + 700 * b], np.int32)
        batches.append((torch.from_numpy(clips).pin_memory(), torch.from_numpy(lens).pin_memory()))
    want = []
    for clips, lens in batches:
        out, ol = eng.run(clips.cuda(), lens.cuda())
        want.append((out.clone(), ol.clone()))
    eng.stage(*batches[0], 0)
    for b in range(3):
        if b + 1 < 3:
            eng.stage(*batches[b + 1], (b + 1) & 1)
        out, ol = eng.run_staged(b & 1)
        torch.cuda.synchronize()
        assert torch.equal(ol, want[b][1])
        assert torch.equal(out, want[b][0])


def test_engine_ragged_buckets_match_the_padded_step(pkg):
    """run_ragged (clips ordered by length, every bucket at its own frame count) returns, clip for clip, what the
    padded fixed-shape step returns: frame counts exactly, encoder rows to 1e-5 (the attention kernel walks a
    different number of key tiles, everything else is row-wise)."""
    n = 16000 * 4
    kinds = ["bursts", "gauss", "chirp", "uniform", "bench", "gauss", "chirp"]
    clips = np.stack([fo.synth_pcm(k, n, 60 + i) for i, k in enumerate(kinds)])
    host_lens = [64000, 50000, 16000, 700, 33333, 399, 64000]
    lens = dev(np.array(host_lens, np.int32))
    sd = so.make_state_dict(so.SanmConfig(num_blocks=2), seed=3)
    cfg = pkg.default_config()
    cfg.enc_num_blocks = 2
    eng = pkg.AsrEngine(batch=7, max_samples=n, cfg=cfg, state_dict=sd, n_streams=2, n_buckets=3)
    ref, ref_lens = eng.run(dev(clips), lens)
    ref, ref_lens = ref.clone(), ref_lens.clone()
    order, parts = eng.run_ragged(dev(clips), lens, host_lens)
    torch.cuda.synchronize()
    assert sorted(order) == list(range(7))
    assert [host_lens[i] for i in order] == sorted(host_lens)
    seen = 0
    for lo, hi, out, out_lens in parts:
        assert out.shape[0] == hi - lo and out.shape[2] == 512
        assert out.shape[1] == max(1, max(int(ref_lens[order[j]]) for j in range(lo, hi)))
        for j in range(lo, hi):
            u = order[j]
            t = int(ref_lens[u])
            assert int(out_lens[j - lo]) == t
            assert float((out[j - lo, :t] - ref[u, :t]).abs().max()) <= 1e-5 if t else True
            assert float(out[j - lo, t:].abs().max()) == 0.0 if t < out.shape[1] else True
            seen += 1
    assert seen == 7


# ---- PCM ingress (streaming-session buffer -> pipeline hand-off) -------------------------------------
def _session_floats(n, seed):
    """Float32 session samples with everything the hand-off has to get right: overshoot beyond +-1, exact +-1, values
    whose product with 32767 sits next to an integer, zeros and tiny values of both signs."""
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 0.45, n).astype(np.float32)
    k = np.arange(-32767, 32768, dtype=np.float64)
    near = np.concatenate([np.nextafter((k / 32767.0).astype(np.float32), np.float32(2)),
                           np.nextafter((k / 32767.0).astype(np.float32), np.float32(-2)),
                           (k / 32767.0).astype(np.float32)])
    pos = rng.choice(n, size=min(n // 2, near.size), replace=False)
    x[pos] = rng.permutation(near)[: pos.size]
    x[:8] = [1.0, -1.0, 1.5, -3.0, 0.0, -0.0, 1e-9, -1e-9]
    return x


def test_wire_int16_on_device_is_bit_exact(pkg):
    from oracle import ingress_oracle as ing
    x = _session_floats(400000, 1)
    got = pkg.wire_int16(dev(x))
    assert got.dtype == torch.int16 and got.is_cuda
    assert np.array_equal(got.cpu().numpy(), ing.wire_int16(x))


def test_session_float_frontend_equals_the_int16_hand_off(pkg):
    """float32 session buffer with the hand-off applied inside the fbank kernel == the int16 signal the reference
    writes for its pipeline, fed as int16: identical bits (the kernels differ only in the load), vector and scalar
    load paths, ragged lengths; and <= 1e-3 against the oracle chain."""
    from oracle import ingress_oracle as ing
    fe16 = pkg.WavFrontend()
    fes = pkg.WavFrontend(session_float=True)
    n = 16000 * 2
    x = np.stack([_session_floats(n, 2), _session_floats(n, 3), 0.7 * _session_floats(n, 4)])
    lens = [n, 23457, 401]
    w = ing.wire_int16(x)
    a, an = fe16.forward_fbank(dev(w), lens)
    b, bn = fes.forward_fbank(dev(x), lens)
    assert torch.equal(an, bn) and torch.equal(a, b)
    want, wn = fo.fbank_batch(w, lens)
    assert bn.cpu().tolist() == wn.tolist()
    assert np.max(np.abs(b.cpu().numpy() - want)) <= FBANK_TOL
    fa, la = fe16.forward(dev(w), lens)
    fb, lb = fes.forward(dev(x), lens)
    assert torch.equal(la, lb) and torch.equal(fa, fb)
    # rows that are not 16-byte aligned take the per-sample loads
    odd = np.zeros((2, 4003), np.float32)
    odd[0, :4001] = x[0, :4001]
    odd[1, :3000] = x[1, :3000]
    c, _ = fes.forward_fbank(dev(odd), [4001, 3000])
    d, _ = fe16.forward_fbank(dev(ing.wire_int16(odd)), [4001, 3000])
    assert torch.equal(c, d)
    # and it is not the FunASR float convention (x 32768, no clipping)
    e, _ = pkg.WavFrontend().forward_fbank(dev(x), lens)
    assert not torch.equal(e, b)


def test_streaming_chunks_to_encoder_match_the_reference_hand_off(pkg):
    """The reference's streaming flow on one utterance: base64 WAV chunks of 400 ms -> float32 session buffer ->
    (finalize) int16 hand-off -> recognition.  Here: decode_wav_base64 -> PcmAccumulator on the device -> engine with
    session_float; compared with the oracle chain run on the oracle's int16 hand-off."""
    import base64
    import io
    import wave
    from oracle import ingress_oracle as ing

    def chunk_b64(samples, rate, channels):
        buf = io.BytesIO()
        with wave.open(buf, "wb") as wf:
            wf.setnchannels(channels)
            wf.setsampwidth(2)
            wf.setframerate(rate)
            wf.writeframes(samples.tobytes())
        return base64.b64encode(buf.getvalue()).decode()

    pcm = fo.synth_pcm("bursts", 16000 * 3, 91)
    payloads = [chunk_b64(pcm[i:i + 6400], 16000, 1) for i in range(0, 44800, 6400)]
    payloads.append(chunk_b64(np.repeat(pcm[44800:], 2), 16000, 2))       # a stereo chunk (both channels equal)
    acc = pkg.PcmAccumulator(capacity=10000)
    ref_chunks = []
    for p in payloads:
        got, sr = pkg.decode_wav_base64(p)
        want, _ = ing.decode_wav_base64(p)
        assert sr == 16000 and np.array_equal(got, want)
        acc.append(got)
        ref_chunks.append(want)
    session = ing.accumulate(ref_chunks)
    assert len(acc) == session.size == 48000
    assert np.array_equal(acc.samples().cpu().numpy(), session)
    wire = ing.wire_int16(session)
    pcm16, lens = acc.wire()
    assert np.array_equal(pcm16.cpu().numpy()[0], wire) and lens.cpu().tolist() == [48000]

    rng = np.random.default_rng(6)
    shift = -rng.uniform(6, 14, 560).astype(np.float32)
    scale = (1.0 / rng.uniform(1.5, 4.5, 560)).astype(np.float32)
    cfg_o = so.SanmConfig(num_blocks=2)
    sd = so.make_state_dict(cfg_o, seed=4)
    cfg = pkg.default_config()
    cfg.enc_num_blocks = 2
    eng = pkg.AsrEngine(batch=1, max_samples=48000, cfg=cfg, cmvn=(shift, scale), state_dict=sd, session_float=True)
    out, olens = eng.run(acc.samples().reshape(1, -1), lens)
    out, olens = out.clone(), olens.clone()
    out16, olens16 = eng.run(pcm16, lens)              # the int16 hand-off itself: same bits
    torch.cuda.synchronize()
    assert torch.equal(olens, olens16) and torch.equal(out, out16)
    wf_, wl = fo.frontend(wire[None], [48000], shift, scale)
    assert olens.cpu().tolist() == wl.tolist()
    want, _ = so.encoder_forward(torch.from_numpy(wf_).float(), torch.from_numpy(wl), sd, cfg_o)
    assert float((out.cpu() - want).abs().max()) <= ENC_TOL


def test_streaming_session_on_the_device_holds_what_the_reference_would_send(pkg):
    """The session's accumulation policy (energy gate, 400 ms of pre-speech, two trailing silent chunks) with the
    utterance on the device: after every chunk the device buffer equals the reference policy's numpy buffer, and the
    features of the float buffer (hand-off inside the kernel) equal those of the oracle's int16 hand-off."""
    from oracle import ingress_oracle as ing
    rng = np.random.default_rng(12)
    ref = ing.SessionBuffer(now=50.0)
    ses = pkg.StreamingPcmSession(capacity=5000, now=50.0)
    t = 50.0
    levels = [0.002, 0.003, 0.25, 0.3, 0.002, 0.2, 0.001, 0.001, 0.001, 0.001, 0.001, 0.001]
    for lv in levels:
        x = rng.normal(0, lv, 6400).astype(np.float32)
        t += 0.4
        assert ses.process_chunk(x, t) == ref.process_chunk(x, t)
        assert np.array_equal(ses.samples().cpu().numpy(), ref.audio_buffer)
    assert ses.process_chunk(np.zeros(6400, np.float32), t + 2.0)["should_finalize"]
    assert len(ses.audio) == 7 * 6400          # 1 pre-speech + 2 speech + 1 silent + 1 speech + 2 silent chunks
    n = len(ses.audio)
    fes = pkg.WavFrontend(session_float=True)
    a, an = fes.forward_fbank(ses.samples().reshape(1, -1), [n])
    b, bn = pkg.WavFrontend().forward_fbank(dev(ing.wire_int16(ref.audio_buffer)[None]), [n])
    assert torch.equal(an, bn) and torch.equal(a, b)


def test_http_route_runs_the_engine_on_the_posted_clip(pkg):
    """POST /api/v1/voice/asr with a 16-bit WAV: the encoder states behind the response are those of the engine run on
    the clip's int16 samples (the route decodes to float / 32768 and the kernel scales back: no value changes)."""
    import base64
    import importlib
    import io
    import wave
    from fastapi.testclient import TestClient
    svc = importlib.import_module("toolbox_for_asr_and_tts_amd.service")
    pcm = fo.synth_pcm("bursts", 16000 * 2, 17)
    buf = io.BytesIO()
    with wave.open(buf, "wb") as wf:
        wf.setnchannels(1)
        wf.setsampwidth(2)
        wf.setframerate(16000)
        wf.writeframes(pcm.tobytes())
    cfg = pkg.default_config()
    cfg.enc_num_blocks = 2
    sd = so.make_state_dict(so.SanmConfig(num_blocks=2), seed=8)
    eng = pkg.AsrEngine(batch=1, max_samples=16000 * 5, cfg=cfg, state_dict=sd)
    want, wl = eng.run(dev(pcm[None]), dev(np.array([pcm.size], np.int32)))
    want, wl = want.clone(), wl.clone()
    seen = {}

    def decoder(out, out_lens):
        seen["out"], seen["lens"] = out.clone(), out_lens.clone()
        return "ok"

    client = TestClient(svc.create_app(eng, decoder))
    r = client.post("/api/v1/voice/asr", json={"audio_data": base64.b64encode(buf.getvalue()).decode(),
                                               "diagnosis_session_id": "t"})
    assert r.status_code == 200 and r.json() == {"text": "ok", "success": True, "message": None, "sample_id": None}
    t = int(wl[0])
    assert torch.equal(seen["lens"], wl) and torch.equal(seen["out"][:, :t], want[:, :t])


@pytest.mark.timeout(90)
def test_websocket_session_finalizes_through_the_engine(pkg):
    """Chunks over the WebSocket route -> session policy -> (2 s of silence) -> the utterance held on the device runs
    through an engine built with session_float: the encoder states the decoder hook receives equal those of the engine
    run on the oracle's int16 hand-off of the oracle session's buffer."""
    import base64
    import importlib
    import io
    import json
    import wave
    from fastapi.testclient import TestClient
    from oracle import ingress_oracle as ing
    svc = importlib.import_module("toolbox_for_asr_and_tts_amd.service")

    def b64(x):
        buf = io.BytesIO()
        with wave.open(buf, "wb") as wf:
            wf.setnchannels(1)
            wf.setsampwidth(2)
            wf.setframerate(16000)
            wf.writeframes(x.tobytes())
        return base64.b64encode(buf.getvalue()).decode()

    cfg = pkg.default_config()
    cfg.enc_num_blocks = 2
    sd = so.make_state_dict(so.SanmConfig(num_blocks=2), seed=9)
    eng = pkg.AsrEngine(batch=1, max_samples=16000 * 10, cfg=cfg, state_dict=sd, session_float=True)
    seen = {}

    def decoder(out, out_lens):
        seen["out"], seen["lens"] = out.clone(), out_lens.clone()
        return "text"

    t = [10.0]
    client = TestClient(svc.create_app(eng, decoder, clock=lambda: t[0]))
    ref = ing.SessionBuffer(now=10.0)
    rng = np.random.default_rng(2)
    with client.websocket_connect("/api/v1/voice/asr/ws") as ws:
        assert ws.receive_json()["type"] == "welcome"
        done = False
        for lv in [20, 5000, 8000, 6000, 20, 20, 20, 20, 20, 20]:
            t[0] += 0.4
            x = rng.normal(0, lv, 6400).astype(np.int16)
            want = ref.process_chunk((x / 32768.0).astype(np.float32), t[0])
            ws.send_text(json.dumps({"wav_base64": b64(x)}))
            if want["should_finalize"]:
                assert ws.receive_json()["status"] == "finalizing"
                assert ws.receive_json() == {"type": "result", "status": "completed", "text": "text", "success": True}
                done = True
                break
            ws.send_text("sync")  # answered after the chunk: the shared clock may move on
            assert ws.receive_json()["code"] == "INVALID_JSON"
        assert done
    wire = ing.wire_int16(ref.audio_buffer)
    eng16 = pkg.AsrEngine(batch=1, max_samples=16000 * 10, cfg=cfg, state_dict=sd)
    want, wl = eng16.run(dev(wire[None]), dev(np.array([wire.size], np.int32)))
    torch.cuda.synchronize()
    n = int(wl[0])
    assert torch.equal(seen["lens"].cpu(), wl.cpu()) and torch.equal(seen["out"][:, :n].cpu(), want[:, :n].cpu())


@pytest.mark.parametrize("height", [1, 2])
def test_gemm_kernel_both_tile_heights_vs_fp64(pkg, height):
    """The general GEMM (debug hook) with its tile height forced (flags bits 8-9: 1 = 256 rows, 2 = 128 rows; unforced,
    the launch takes 128 rows where 256 would leave CUs idle): every epilogue, row counts that end inside a tile and
    inside a wave's share, K from one pair of K-tiles up, against an fp64 product of the same fp16 operands."""
    import ctypes as C

    ctx = pkg.SANMEncoder(num_blocks=1).ctx
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    gen = torch.Generator().manual_seed(60 + height)
    for (m, n, k, epi) in [(1, 256, 128, 0), (129, 256, 128, 1), (300, 512, 256, 2), (513, 512, 384, 1), (1000, 768, 640, 2),
                           (257, 256, 2048, 0), (4100, 512, 512, 2)]:
        a = (torch.randn(m, k, generator=gen) * 0.5).half().cuda()
        w = (torch.randn(n, k, generator=gen) * 0.05).half().cuda()
        bias = torch.randn(n, generator=gen).cuda()
        add16 = torch.randn(m, n, generator=gen).half().cuda()
        add32 = torch.randn(m, n, generator=gen).cuda()
        guard = 64  # rows behind the output keep their fill
        out = torch.full((m + guard, n), 7.0, device="cuda", dtype=torch.float32 if epi == 2 else torch.float16)
        outs = []
        for flags in (height << 8, 0):
            out[:m] = 0
            ctx.check(ctx.lib.asrfe_debug_gemm(ctx.handle, a.data_ptr(), k, w.data_ptr(), k, bias.data_ptr(), m, n, k,
                                               out.data_ptr(), n, add16.data_ptr() if epi == 2 else None,
                                               add32.data_ptr() if epi == 2 else None, epi, flags, stream))
            torch.cuda.synchronize()
            assert bool((out[m:] == 7.0).all()), (m, n, k, epi, flags)
            outs.append(out[:m].clone())
        ref = a.double() @ w.double().T + bias.double()
        if epi == 1:
            ref = torch.relu(ref)
        if epi == 2:
            ref = ref + add16.double() + add32.double()
        for o in outs:
            err = (o.double() - ref).abs().max().item()
            assert err < (2e-5 if epi == 2 else 4e-3), (m, n, k, epi, err)
        # the K loop visits the same k in the same order at either height: the unforced launch gives the same bits
        assert torch.equal(outs[0], outs[1])


def test_attention_fsmn_kernel_vs_fp64(pkg):
    """The attention + FSMN kernel on its own (debug hook) against an fp64 evaluation of the same 16-bit inputs:
    ragged lengths, a length below the FSMN half-width, T' around the 64-key and 128-query tile edges."""
    import ctypes as C
    cfg = so.SanmConfig(num_blocks=1)
    sd = so.make_state_dict(cfg, seed=2, dtype=torch.float32)
    gen = torch.Generator().manual_seed(3)
    sd["encoders0.0.self_attn.fsmn_block.weight"] = torch.randn(512, 1, 11, generator=gen) * 0.2
    enc = pkg.SANMEncoder(num_blocks=1)
    enc.load_state_dict(sd)
    ctx = enc.ctx
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    w = sd["encoders0.0.self_attn.fsmn_block.weight"].double().cuda()
    for T, ll in [(1, [1]), (7, [7, 3]), (64, [64, 5]), (65, [65, 64, 2]), (129, [129, 128, 127]), (256, [256, 1, 0]),
                  (257, [257, 193]), (300, [300, 191, 64, 6]), (513, [513, 449, 65])]:
        B = len(ll)
        qkv = (torch.randn(B, T, 1536, generator=gen) * 1.5).half().cuda()
        lens = torch.tensor(ll, dtype=torch.int32, device="cuda")
        att = torch.full((B, T, 512), float("nan"), dtype=torch.float16, device="cuda")
        fsmn = torch.full((B, T, 512), float("nan"), dtype=torch.float16, device="cuda")
        ctx.check(ctx.lib.asrfe_debug_attention(ctx.handle, 0, qkv.data_ptr(), lens.data_ptr(), B, T, att.data_ptr(),
                                                fsmn.data_ptr(), 0, stream))
        q, k, v = [t.double() for t in qkv.split(512, dim=-1)]
        mask = torch.arange(T, device="cuda")[None, :] < lens[:, None]
        ref = torch.zeros(B, T, 512, dtype=torch.float64, device="cuda")
        for hd in range(4):
            sl = slice(128 * hd, 128 * hd + 128)
            s = (q[..., sl] @ k[..., sl].transpose(1, 2) / math.sqrt(128.0)).masked_fill(~mask[:, None, :], float("-inf"))
            ref[..., sl] = torch.softmax(s, dim=-1) @ v[..., sl]
        vm = v * mask[..., None]
        fref = (torch.nn.functional.conv1d(torch.nn.functional.pad(vm.transpose(1, 2), (5, 5)), w, groups=512).transpose(1, 2)
                + vm) * mask[..., None]
        assert torch.isfinite(att.float()).all() and torch.isfinite(fsmn.float()).all()  # padded rows too: they feed later layers
        # (an utterance without keys has no defined attention output: only finiteness is asked of its rows)
        err = torch.where(mask[..., None], att.double() - ref, torch.zeros_like(ref))
        assert float(err.abs().max()) <= 4e-3, (T, ll)
        assert float((fsmn.double() - fref).abs().max()) <= 8e-3, (T, ll)
        assert float((fsmn.float() * (~mask[..., None])).abs().max()) == 0.0


def test_engine_step_replays_from_a_hip_graph(pkg):
    """The compute calls only enqueue kernels on the caller's stream (no allocation, no synchronisation), so a
    step can be captured into a HIP graph and replayed; the replay reproduces the eager result bit for bit."""
    n = 16000 * 3
    clips = np.stack([fo.synth_pcm(k, n, 90 + i) for i, k in enumerate(["gauss", "chirp", "bursts"])])
    lens = dev(np.array([48000, 30000, 20000], np.int32))
    sd = so.make_state_dict(so.SanmConfig(num_blocks=2), seed=4)
    cfg = pkg.default_config()
    cfg.enc_num_blocks = 2
    eng = pkg.AsrEngine(batch=3, max_samples=n, cfg=cfg, state_dict=sd, n_streams=2)
    pcm = dev(clips)
    want = eng.run(pcm, lens)[0].clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        eng.run(pcm, lens)
    eng.out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(eng.out, want)


def test_latency_plan_row_limit_and_tile_edges(pkg):
    """The latency plan at its edges: one row, a row count that is no multiple of 32, exactly the limit (latency plan)
    and one clip more (throughput plan); every case against the oracle, 3 layers with trained-like LayerNorm parameters."""
    cfg = so.SanmConfig(num_blocks=3)
    sd = _trained_like(so.make_state_dict(cfg, seed=40, dtype=torch.float32), seed=41)
    enc = pkg.SANMEncoder(num_blocks=3, small_batch_rows=256)
    enc.load_state_dict(sd)
    for b, tp, lens in [(1, 1, [1]), (1, 50, [50]), (3, 37, [37, 5, 36]), (2, 128, [128, 100]), (2, 129, [129, 64])]:
        feats = so.synth_feats(b, tp, 560, seed=tp)
        lt = torch.tensor(lens, dtype=torch.int32)
        out, _, _ = enc.forward(feats.cuda(), lt)
        want, _ = so.encoder_forward(feats.double(), lt, {k: v.double() for k, v in sd.items()}, cfg)
        assert float((out.cpu().double() - want).abs().max()) <= ENC_TOL, (b, tp)


def test_calls_of_different_sizes_on_one_context_do_not_disturb_each_other(pkg):
    """One context serves a large engine (throughput plan), a one-utterance engine (latency plan), the head and the
    allocation-free entry point in turn: every repeat gives the bits of the first call, whatever ran in between."""
    from toolbox_for_asr_and_tts_amd.decoder import ParaformerHead
    from oracle import paraformer_decoder_oracle as po
    nb = 3
    cfg = pkg.default_config()
    cfg.enc_num_blocks = nb
    big = pkg.AsrEngine(batch=40, max_samples=64000, cfg=cfg, state_dict=so.make_state_dict(so.SanmConfig(num_blocks=nb), seed=41))
    one = pkg.AsrEngine(batch=1, max_samples=40000, ctx=big.ctx)
    dcfg = po.DecoderConfig(num_blocks=2, vocab_size=300)
    head = ParaformerHead(big.ctx, num_blocks=2, vocab_size=300)
    head.load_state_dict(po.make_state_dict(dcfg, seed=42))
    rng = np.random.default_rng(43)
    pcm_b = dev(rng.integers(-3000, 3000, size=(40, 64000)).astype(np.int16))
    len_b = torch.tensor(rng.integers(1000, 64000, size=40).astype(np.int32), device="cuda")
    pcm_1 = dev(rng.integers(-3000, 3000, size=(1, 40000)).astype(np.int16))
    len_1 = torch.tensor([40000], dtype=torch.int32, device="cuda")

    def all_calls():
        ob = big.run(pcm_b, len_b)[0].clone()
        ids_b = head.forward(big.out, big.feat_lens)["token_ids"].clone()
        o1 = one.run(pcm_1, len_1)[0].clone()
        ids_1 = head.forward(one.out, one.feat_lens)["token_ids"].clone()
        return ob, ids_b, o1, ids_1

    first = all_calls()
    for _ in range(2):
        again = all_calls()
        for a, b in zip(first, again):
            assert torch.equal(a, b)
    assert first[0].shape[0] == 40 and big.tp_max * 40 > 2048 and one.tp_max <= 2048  # the two plans were both taken
