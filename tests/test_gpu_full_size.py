"""GPU parity at the benchmark's own sizes, the fp16 range policy and the truncation rules of the frontend.

The full-size fixtures (tests/golden/encoder_full_50l_t500.npz, bench_clips_50l.npz) come from the fp64 oracle chain
(tests/golden/make_golden.py full).  Tolerances: frame counts exact, features <= 1e-3, encoder output <= 1e-2.
"""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

from oracle import fbank_oracle as fo
from oracle import sanm_oracle as so

pytestmark = pytest.mark.gpu

FBANK_TOL = 1e-3
ENC_TOL = 1e-2
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
SAMPLES = 480000


@pytest.fixture(scope="module")
def pkg():
    import toolbox_for_asr_and_tts_amd as p

    p.load()
    return p


def dev(x):
    return torch.as_tensor(x).to("cuda")


# ---- encoder: 50 blocks at T' = 500 -----------------------------------------------------------------------
def test_encoder_50_blocks_full_length_golden(pkg):
    """50 blocks, B = 2, T' = 500 with lens [500, 333] against the fp64 oracle fixture: selected rows to 1e-2, per-clip
    column sums, mean magnitude, zero rows past the length."""
    g = np.load(os.path.join(GOLDEN, "encoder_full_50l_t500.npz"))
    cfg = so.SanmConfig(num_blocks=int(g["num_blocks"]))
    sd = so.make_state_dict(cfg, seed=int(g["seed"]), dtype=torch.float32)
    feats = so.synth_feats(2, 500, 560, seed=int(g["feats_seed"]))
    lens = torch.from_numpy(g["lens"])
    enc = pkg.SANMEncoder(num_blocks=cfg.num_blocks)
    enc.load_state_dict(sd)
    out, olens, _ = enc.forward(feats.cuda(), lens)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert olens.cpu().tolist() == g["lens"].tolist()
    assert np.isfinite(got).all()
    e0 = np.max(np.abs(got[0, g["rows"]] - g["rows_b0"]))
    e1 = np.max(np.abs(got[1, g["rows1"]] - g["rows_b1"]))
    assert max(e0, e1) <= ENC_TOL, (e0, e1)
    # a column sum over n rows may drift by n times the row tolerance at most; in practice it stays far below
    assert np.max(np.abs(got.sum(axis=1) - g["col_sums"])) <= ENC_TOL * 50
    assert abs(np.abs(got).mean() - float(g["abs_mean"])) <= 1e-3
    assert float(np.abs(got[1, 333:]).max()) == 0.0


# ---- the benchmark batch itself: 64 x 30 s through the engine --------------------------------------------
def test_bench_batch_64x30s_matches_oracle_fixture(pkg):
    """The step bench.py times (AsrEngine, batch 64, 480000 samples per clip, 50 blocks, the benchmark's PCM and
    weights, two streams): clips 0, 31 and 63 against the fp64 oracle chain (features <= 1e-3, encoder rows <= 1e-2,
    column sums), every output finite, and the same three clips run as a batch of 3 give the same rows (<= 1e-5 under
    the same launch plan, within the encoder tolerance under the latency plan a call of 1500 rows takes by default):
    a clip's result does not depend on the batch around it, so the fixture speaks for all 64."""
    from toolbox_for_asr_and_tts_amd.synth import bench_pcm, encoder_state_dict
    g = np.load(os.path.join(GOLDEN, "bench_clips_50l.npz"))
    clips = g["clips"].tolist()
    rows = g["rows"]
    sd = encoder_state_dict(num_blocks=50, seed=int(g["weights_seed"]))
    pcm = bench_pcm(64, SAMPLES, seed=int(g["pcm_seed"]))
    eng = pkg.AsrEngine(batch=64, max_samples=SAMPLES, state_dict=sd, n_streams=2)
    lens = torch.full((64,), SAMPLES, dtype=torch.int32, device="cuda")
    out, olens = eng.run(dev(pcm), lens)
    torch.cuda.synchronize()
    assert olens.cpu().tolist() == [500] * 64 and g["olens"].tolist() == [500] * 3
    assert bool(torch.isfinite(out).all())
    feats = eng.feats[clips].cpu().numpy()
    assert np.max(np.abs(feats[:, rows] - g["feat_rows"])) <= FBANK_TOL
    assert np.max(np.abs(feats.sum(axis=1) - g["feat_col_sums"])) <= FBANK_TOL * 500
    got = out[clips].cpu().numpy()
    err = np.max(np.abs(got[:, rows] - g["out_rows"]))
    assert err <= ENC_TOL, err
    # column sums over the 500 rows: the clips' zero prefix makes ~50 rows identical, so their rounding errors add up
    # instead of averaging out; the mean error per element must still stay well inside the row tolerance
    assert np.max(np.abs(got.sum(axis=1) - g["out_col_sums"])) / 500 <= ENC_TOL / 4
    # single stream, same engine: bit-identical
    out1 = eng.run_single_stream(dev(pcm), lens)[0]
    torch.cuda.synchronize()
    assert torch.equal(out1[clips].cpu(), torch.from_numpy(got))
    # and the step repeats bit for bit, all 32000 rows of it (the post-attention kernel counts its loads in flight by
    # hand: a count that is too loose somewhere would show as rows that differ from run to run)
    first = out.clone()
    for _ in range(12):
        again = eng.run(dev(pcm), lens)[0]
        torch.cuda.synchronize()
        assert torch.equal(again, first)
    # the three clips alone, under the same (throughput) launch plan: the same rows
    from toolbox_for_asr_and_tts_amd import _lib
    cfg0 = _lib.default_config()
    cfg0.enc_small_batch_rows = 0
    eng3 = pkg.AsrEngine(batch=3, max_samples=SAMPLES, cfg=cfg0, state_dict=sd)
    out3, _ = eng3.run(dev(pcm[clips]), lens[:3])
    torch.cuda.synchronize()
    assert float((out3.cpu() - torch.from_numpy(got)).abs().max()) <= 1e-5
    # and as the library runs 1500 rows by default (the latency plan: another summation order), against the fixture
    eng3l = pkg.AsrEngine(batch=3, max_samples=SAMPLES, state_dict=sd)
    out3l = eng3l.run(dev(pcm[clips]), lens[:3])[0].cpu().numpy()
    assert np.max(np.abs(out3l[:, rows] - g["out_rows"])) <= ENC_TOL
    assert np.max(np.abs(out3l - got)) <= ENC_TOL


@pytest.mark.parametrize("plan", ["ASRFE_UNFUSED_FFN", "ASRFE_SEPARATE_QKV"])
def test_two_streams_repeat_bit_for_bit_under_the_plans_with_a_gemm_per_layer(pkg, monkeypatch, plan):
    """The plans that launch the general GEMM in every layer (four products per layer unfused, one with a separate QKV
    projection), two chunks on two streams, 64 x 30 s: every repetition gives the same bits.  The GEMM refills a
    half-tile of its LDS ring right behind a workgroup barrier, and its operands are then lines the previous kernel
    has just written (hot in L2, back within a few hundred cycles): a fragment read left in flight across that barrier
    shows as blocks of 8 to 32 rows that change from run to run (it did, several times per run of this test, before
    the kernel drained its reads in front of the barrier)."""
    from toolbox_for_asr_and_tts_amd.synth import bench_pcm, encoder_state_dict
    from toolbox_for_asr_and_tts_amd import _lib
    monkeypatch.setenv(plan, "1")
    n_blocks = 16
    sd = encoder_state_dict(num_blocks=n_blocks, seed=0)
    cfg = _lib.default_config()
    cfg.enc_num_blocks = n_blocks
    pcm = dev(bench_pcm(64, SAMPLES, seed=1234))
    lens = torch.full((64,), SAMPLES, dtype=torch.int32, device="cuda")
    eng = pkg.AsrEngine(batch=64, max_samples=SAMPLES, cfg=cfg, state_dict=sd, n_streams=2)
    first = eng.run(pcm, lens)[0].clone()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(first).all())
    for it in range(8):
        again = eng.run(pcm, lens)[0]
        torch.cuda.synchronize()
        if not torch.equal(again, first):
            utts = (again != first).any(dim=2).any(dim=1).nonzero().flatten().tolist()
            raise AssertionError(f"{plan}: repetition {it} differs from the first run in utterances {utts}")
    # one stream, same engine: the same bits (the plan is a property of the context, not of the stream count)
    single = eng.run_single_stream(pcm, lens)[0]
    torch.cuda.synchronize()
    assert torch.equal(single, first)


def test_fbank_256x30s(pkg):
    """asrfe_fbank on the fbank-only configuration (256 clips of 30 s): frame counts, finiteness, batch-position
    independence (clips 0, 128, 255 alone give the same bits) and oracle frames at the start, in the middle and at the
    end of those clips."""
    from toolbox_for_asr_and_tts_amd.synth import bench_pcm
    fe = pkg.WavFrontend()
    pcm = bench_pcm(256, SAMPLES, seed=4321)
    x = dev(pcm)
    out, nfr = fe.forward_fbank(x, [SAMPLES] * 256)
    torch.cuda.synchronize()
    assert out.shape == (256, 2998, 80) and nfr.cpu().tolist() == [2998] * 256
    assert bool(torch.isfinite(out).all())
    pick = [0, 128, 255]
    alone, _ = fe.forward_fbank(x[pick].contiguous(), [SAMPLES] * 3)
    assert torch.equal(alone, out[pick])
    for b in pick:
        for t0, n in [(0, 4), (296, 8), (1500, 8), (2990, 8)]:
            seg = pcm[b, 160 * t0: 160 * (t0 + n - 1) + 400]
            want = fo.kaldi_fbank(seg)
            assert want.shape[0] == n
            assert np.max(np.abs(out[b, t0:t0 + n].cpu().numpy() - want)) <= FBANK_TOL, (b, t0)
    # checksum of checksums: summing the per-clip sums equals the sum over the whole tensor (no clip lost or doubled)
    per_clip = out.double().sum(dim=(1, 2))
    assert abs(float(per_clip.sum()) - float(out.double().sum())) <= 1e-6 * abs(float(per_clip.sum()))
    # the 10 % zero prefix of every clip sits on the log floor
    zp = out[:, : (SAMPLES // 10 - 400) // 160]
    assert torch.all(zp == zp[0, 0, 0]) and abs(float(zp[0, 0, 0]) - math.log(fo.LOG_FLOOR)) < 1e-5


# ---- fp16 range ------------------------------------------------------------------------------------------
def _scaled_ffn(sd, factor):
    """w_1, b_1 scaled up and w_2 scaled down by the same factor: the function is unchanged (ReLU is positively
    homogeneous) while the hidden activations grow by the factor."""
    out = dict(sd)
    for k, v in sd.items():
        if k.endswith("feed_forward.w_1.weight") or k.endswith("feed_forward.w_1.bias"):
            out[k] = v * factor
        elif k.endswith("feed_forward.w_2.weight"):
            out[k] = v / factor
    return out


def test_fp16_operands_hold_large_hidden_activations(pkg):
    """Hidden activations in the thousands (w_1 x 2048: the weight-norm bound is 2.1e4, still inside the fp16 range):
    the load is accepted and the output stays within 1e-2 of the fp64 oracle."""
    cfg = so.SanmConfig(num_blocks=3)
    sd = _scaled_ffn(so.make_state_dict(cfg, seed=31, dtype=torch.float32), 2048.0)
    feats = so.synth_feats(2, 130, 560, seed=32)
    lens = torch.tensor([130, 71], dtype=torch.int32)
    enc = pkg.SANMEncoder(num_blocks=3)
    enc.load_state_dict(sd)
    out, _, _ = enc.forward(feats.cuda(), lens)
    want, _ = so.encoder_forward(feats.double(), lens, {k: v.double() for k, v in sd.items()}, cfg)
    bound = float(sd["encoders.0.feed_forward.w_1.weight"].double().norm(dim=1).max()) * math.sqrt(512.0)
    assert 1e4 < bound < 65504.0  # what the load call proves: hidden values cannot leave the fp16 range
    assert bool(torch.isfinite(out).all())
    err = float((out.cpu().double() - want).abs().max())
    assert err <= ENC_TOL, err


def test_hidden_units_beyond_the_fp16_range_are_rescaled_exactly(pkg):
    """w_1 x 65536 puts the proven bound of the hidden layer (||w||_2 sqrt(512) + |b|) beyond 65504.  The fp16 context
    moves a power of two from each such unit's row of w_1 into its column of w_2 while loading (the function is
    unchanged, nothing is rounded), so the load is accepted and the output stays within 1e-2 of the fp64 oracle."""
    cfg = so.SanmConfig(num_blocks=2)
    sd = _scaled_ffn(so.make_state_dict(cfg, seed=33, dtype=torch.float32), 65536.0)
    bound = float(sd["encoders.0.feed_forward.w_1.weight"].double().norm(dim=1).max()) * math.sqrt(512.0)
    assert bound > 65504.0
    feats = so.synth_feats(2, 130, 560, seed=34)
    lens = torch.tensor([130, 71], dtype=torch.int32)
    enc = pkg.SANMEncoder(num_blocks=2)
    enc.load_state_dict(sd)
    out, _, _ = enc.forward(feats.cuda(), lens)
    want, _ = so.encoder_forward(feats.double(), lens, {k: v.double() for k, v in sd.items()}, cfg)
    assert bool(torch.isfinite(out).all())
    err = float((out.cpu().double() - want).abs().max())
    assert err <= ENC_TOL, err
    # the throughput plan reads the same rescaled weights from its per-layer stream
    big = so.synth_feats(9, 256, 560, seed=35)
    blens = torch.tensor([256] * 8 + [100], dtype=torch.int32)
    outb, _, _ = enc.forward(big.cuda(), blens)
    wantb, _ = so.encoder_forward(big.double(), blens, {k: v.double() for k, v in sd.items()}, cfg)
    errb = float((outb.cpu().double() - wantb).abs().max())
    assert errb <= ENC_TOL, errb


def test_q_beyond_the_fp16_range_is_balanced_against_k(pkg):
    """q rows x 2^18 and k rows x 2^-18: every score q . k is what it was, but q's bound leaves the fp16 range (and k
    sinks towards its subnormals).  The load call moves the power of two back, dimension by dimension; the output stays
    within 1e-2 of the fp64 oracle in both layers' positions (layer 0 has its own 560-wide projection)."""
    cfg = so.SanmConfig(num_blocks=3)
    sd = so.make_state_dict(cfg, seed=36, dtype=torch.float32)
    for layer in ("encoders0.0", "encoders.1"):
        for part in ("weight", "bias"):
            k = f"{layer}.self_attn.linear_q_k_v.{part}"
            v = sd[k].clone()
            v[:512] *= 2.0 ** 18
            v[512:1024] *= 2.0 ** -18
            sd[k] = v
    feats = so.synth_feats(2, 130, 560, seed=37)
    lens = torch.tensor([130, 71], dtype=torch.int32)
    enc = pkg.SANMEncoder(num_blocks=3)
    enc.load_state_dict(sd)
    out, _, _ = enc.forward(feats.cuda(), lens)
    want, _ = so.encoder_forward(feats.double(), lens, {k: v.double() for k, v in sd.items()}, cfg)
    err = float((out.cpu().double() - want).abs().max())
    assert err <= ENC_TOL, err


def test_what_cannot_be_rescaled_is_refused_and_loads_as_bf16(pkg):
    """v feeds the FSMN memory, which joins the residual stream as it is: no partner to take its scale.  v rows x 1e5
    are refused with ASRFE_ERR_RANGE and a message naming the layer; so are q and k when both are beyond the range
    (balancing leaves each at the geometric mean).  A bf16 context takes the same weights."""
    cfg = so.SanmConfig(num_blocks=2)
    sd = so.make_state_dict(cfg, seed=33, dtype=torch.float32)
    w = sd["encoders.0.self_attn.linear_q_k_v.weight"].clone()
    w[1024:] *= 1e5
    sd["encoders.0.self_attn.linear_q_k_v.weight"] = w
    with pytest.raises(pkg.AsrfeError, match=r"layer 1 v \(and the attention output\).*fp16 range") as ei:
        pkg.SANMEncoder(num_blocks=2).load_state_dict(sd)
    assert ei.value.code == -7
    sd2 = so.make_state_dict(cfg, seed=33, dtype=torch.float32)
    w = sd2["encoders.0.self_attn.linear_q_k_v.weight"].clone()
    w[:1024] *= 1e5
    sd2["encoders.0.self_attn.linear_q_k_v.weight"] = w
    with pytest.raises(pkg.AsrfeError, match="layer 1 q / k"):
        pkg.SANMEncoder(num_blocks=2).load_state_dict(sd2)
    encb = pkg.SANMEncoder(num_blocks=2, compute_dtype="bfloat16")
    encb.load_state_dict(sd)
    feats = so.synth_feats(1, 40, 560, seed=34)
    out, _, _ = encb.forward(feats.cuda(), [40])
    assert bool(torch.isfinite(out).all())


def test_fp16_conversions_saturate(pkg):
    """The kernels convert to fp16 with saturation: a GEMM whose results pass 65504 stores +-65504, not infinities
    (debug hook; fp16 context), and values inside the range are untouched."""
    enc = pkg.SANMEncoder(num_blocks=1)
    ctx = enc.ctx
    m, n, k = 256, 256, 128
    a = torch.full((m, k), 100.0, dtype=torch.float16, device="cuda")
    w = torch.zeros((n, k), dtype=torch.float16, device="cuda")
    w[0::3] = 100.0     # 128 * 100 * 100 = 1.28e6 -> +65504
    w[1::3] = -100.0    # -> -65504
    w[2::3] = 0.25      # 128 * 100 * 0.25 = 3200, exact
    bias = torch.zeros(n, dtype=torch.float32, device="cuda")
    out = torch.empty((m, n), dtype=torch.float16, device="cuda")
    ctx.check(ctx.lib.asrfe_debug_gemm(ctx.handle, a.data_ptr(), k, w.data_ptr(), k, bias.data_ptr(), m, n, k,
                                       out.data_ptr(), n, None, None, 0, 0,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    o = out.float()
    assert bool(torch.isfinite(o).all())
    assert torch.all(o[:, 0::3] == 65504.0) and torch.all(o[:, 1::3] == -65504.0) and torch.all(o[:, 2::3] == 3200.0)


# ---- truncation: outputs with fewer rows than the clip has frames ----------------------------------------
@pytest.mark.parametrize("lfr_m,lfr_n", [(7, 6), (5, 1)])
def test_outputs_shorter_than_the_clip_truncate_it(pkg, lfr_m, lfr_n):
    """t_max / tp_max below a clip's frame count: the clip is processed as if cut after frame t_max - 1 (fused:
    tp_max * lfr_n), the reported counts are clamped, nothing is written or read past the given rows (guard rows
    behind the buffers keep their fill), split and fused paths agree with the oracle on the cut clip."""
    fe = pkg.WavFrontend(lfr_m=lfr_m, lfr_n=lfr_n)
    ctx = fe.ctx
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    n = 16000
    clips = np.stack([fo.synth_pcm("gauss", n, 3), fo.synth_pcm("bursts", n, 4), fo.synth_pcm("chirp", n, 5)])
    lens = [16000, 9000, 2000]          # 98, 54 and 11 frames
    x = dev(clips)
    lens_dev = dev(np.array(lens, np.int32))
    t_max = 40
    guard = 7.5
    logmel = torch.full((3 * t_max + 8, 80), guard, dtype=torch.float32, device="cuda")
    nfr = torch.zeros(3, dtype=torch.int32, device="cuda")
    ctx.check(ctx.lib.asrfe_fbank(ctx.handle, x.data_ptr(), 0, x.stride(0), lens_dev.data_ptr(), 3, t_max,
                                  logmel.data_ptr(), nfr.data_ptr(), stream))
    torch.cuda.synchronize()
    assert nfr.cpu().tolist() == [40, 40, 11]
    assert torch.all(logmel[3 * t_max:] == guard)
    lm = logmel[:3 * t_max].reshape(3, t_max, 80).cpu().numpy()
    for b, ln in enumerate(lens):
        t = min(fo.num_frames(ln), t_max)
        want = fo.kaldi_fbank(clips[b, :ln])[:t]
        assert np.max(np.abs(lm[b, :t] - want)) <= FBANK_TOL
        assert np.all(lm[b, t:] == 0.0)
    # split LFR/CMVN with fewer rows than the frames allow
    dim = 80 * lfr_m
    tp_max = 5
    feats = torch.full((3 * tp_max + 4, dim), guard, dtype=torch.float32, device="cuda")
    flens = torch.zeros(3, dtype=torch.int32, device="cuda")
    ctx.check(ctx.lib.asrfe_lfr_cmvn(ctx.handle, logmel.data_ptr(), nfr.data_ptr(), 3, t_max, tp_max, feats.data_ptr(), 0,
                                     flens.data_ptr(), stream))
    torch.cuda.synchronize()
    assert torch.all(feats[3 * tp_max:] == guard)
    f = feats[:3 * tp_max].reshape(3, tp_max, dim).cpu().numpy()
    for b in range(3):
        t = int(nfr[b])
        want = fo.apply_lfr(lm[b, :t].astype(np.float64), lfr_m, lfr_n)
        tp = min(want.shape[0], tp_max)
        assert int(flens[b]) == tp
        assert np.array_equal(f[b, :tp], want[:tp].astype(np.float32))
    # a frame count beyond the rows of logmel (a caller's mistake) is clamped, not followed out of the buffer
    big = torch.tensor([1000, 40, 11], dtype=torch.int32, device="cuda")
    ctx.check(ctx.lib.asrfe_lfr_cmvn(ctx.handle, logmel.data_ptr(), big.data_ptr(), 3, t_max, tp_max, feats.data_ptr(), 0,
                                     flens.data_ptr(), stream))
    torch.cuda.synchronize()
    assert int(flens[0]) == min(fo.num_lfr_frames(40, lfr_n), tp_max)
    # fused path: tp_max rows -> the clip is cut after tp_max * lfr_n frames
    feats2 = torch.full((3 * tp_max + 4, dim), guard, dtype=torch.float32, device="cuda")
    flens2 = torch.zeros(3, dtype=torch.int32, device="cuda")
    ctx.check(ctx.lib.asrfe_frontend(ctx.handle, x.data_ptr(), 0, x.stride(0), lens_dev.data_ptr(), 3, tp_max,
                                     feats2.data_ptr(), 0, flens2.data_ptr(), stream))
    torch.cuda.synchronize()
    assert torch.all(feats2[3 * tp_max:] == guard)
    f2 = feats2[:3 * tp_max].reshape(3, tp_max, dim).cpu().numpy()
    for b, ln in enumerate(lens):
        t = min(fo.num_frames(ln), tp_max * lfr_n)
        want = fo.apply_lfr(fo.kaldi_fbank(clips[b, :ln])[:t], lfr_m, lfr_n)
        tp = want.shape[0]
        assert tp <= tp_max and int(flens2[b]) == tp
        assert np.max(np.abs(f2[b, :tp] - want)) <= FBANK_TOL
        assert np.all(f2[b, tp:] == 0.0)


def test_two_contexts_in_one_process_and_cmvn_reload_between_launches(pkg):
    """Kernel attributes are set per context at creation (not once per process), and asrfe_load_cmvn waits for
    launches in flight before it replaces the table: back-to-back launches with different tables on a side stream
    each see their own."""
    fe_a = pkg.WavFrontend()
    fe_b = pkg.WavFrontend()
    pcm = dev(np.stack([fo.synth_pcm("gauss", 160000, 50 + i) for i in range(8)]))
    lens = [160000] * 8
    side = torch.cuda.Stream()
    outs = []
    with torch.cuda.stream(side):
        for k in range(4):
            fe_a.set_cmvn(np.full(560, -float(k), np.float32), np.full(560, 1.0 + k, np.float32))
            outs.append(fe_a.forward(pcm, lens)[0])
    side.synchronize()
    base, _ = fe_b.forward(pcm, lens)
    torch.cuda.synchronize()
    for k in range(4):
        assert torch.equal(outs[k], (base - float(k)) * (1.0 + k))
