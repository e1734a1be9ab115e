"""FSMN voice-activity path on the GPU, through the C ABI: the network against the fp64 oracle (one shot, batched with
ragged lengths, chunk by chunk with the per-stream cache), the frame level kernel, the streaming frontend, and
``AutoModel(model=<vad directory>).generate(input=chunk, cache=..., is_final=False, chunk_size=400)`` as the reference calls
it per 400 ms chunk (voice_interface.py:1585-1590).  PARITY UNPINNED (see oracle/fsmn_vad_oracle.py): the oracle restates
published FunASR modules; the reference ships no fixture for them.

Tolerance: the products run on fp16 operands (weights and activations rounded to 11 bits, fp32 accumulation), which moves
a class probability by up to a few 1e-3; 5e-3 absolute on probabilities in [0, 1] is the bar written below (bf16: 4e-2)."""
import numpy as np
import pytest
import torch

from oracle import fbank_oracle as fo
from oracle import fsmn_vad_oracle as vo
from tests import vad_util as vu

pytestmark = pytest.mark.gpu

TOL_F16 = 5e-3
TOL_BF16 = 4e-2


@pytest.fixture(scope="module")
def pkg():
    import toolbox_for_asr_and_tts_amd as p

    p.load()
    return p


def make_net(pkg, sd, dtype=None):
    from toolbox_for_asr_and_tts_amd import _lib
    cfg = pkg.default_config()
    cfg.lfr_m, cfg.lfr_n, cfg.enc_input_size = 5, 1, 400
    if dtype is not None:
        cfg.enc_compute_dtype = dtype
    ctx = _lib.Context(cfg)
    return pkg.FsmnVadNet(ctx).load_state_dict(sd)


@pytest.mark.parametrize("t", [1, 19, 137, 1000])
def test_scores_match_the_oracle(pkg, t):
    sd = vo.make_state_dict(seed=11)
    net = make_net(pkg, sd)
    rng = np.random.default_rng(t)
    x = rng.standard_normal((t, 400)).astype(np.float32)
    want, _ = vo.fsmn_forward(sd, x)
    got = net(torch.from_numpy(x).cuda()[None])[0].cpu().numpy()
    assert got.shape == (t, 248)
    assert np.allclose(got.sum(axis=1), 1.0, atol=1e-5)
    assert np.abs(got - want).max() < TOL_F16
    # the distribution is not flat: the comparison means something
    assert want.max(axis=1).mean() > 5.0 / 248


def test_bf16_operands(pkg):
    from toolbox_for_asr_and_tts_amd import _lib
    sd = vo.make_state_dict(seed=12)
    net = make_net(pkg, sd, _lib.DT_BF16)
    x = np.random.default_rng(0).standard_normal((200, 400)).astype(np.float32)
    want, _ = vo.fsmn_forward(sd, x)
    got = net(torch.from_numpy(x).cuda()[None])[0].cpu().numpy()
    assert np.abs(got - want).max() < TOL_BF16


def test_streams_with_caches_and_ragged_chunks(pkg):
    sd = vo.make_state_dict(seed=13)
    net = make_net(pkg, sd)
    rng = np.random.default_rng(5)
    B = 3
    # per stream: the chunk lengths of four calls (0 = the stream brings nothing to that call)
    chunk_lens = [[40, 7, 0, 25], [3, 40, 19, 1], [33, 0, 40, 40]]
    total = [sum(c) for c in chunk_lens]
    xs = [rng.standard_normal((n, 400)).astype(np.float32) for n in total]
    want = [vo.fsmn_forward(sd, x) for x in xs]
    cache = net.new_cache(B, "cuda")
    pos = [0] * B
    got = [[] for _ in range(B)]
    for call in range(4):
        lens = [chunk_lens[b][call] for b in range(B)]
        T = 40
        feats = np.zeros((B, T, 400), dtype=np.float32)
        for b in range(B):
            feats[b, :lens[b]] = xs[b][pos[b]:pos[b] + lens[b]]
            feats[b, lens[b]:] = 77.0                     # rows past a stream's length must not reach its cache
        s = net(torch.from_numpy(feats).cuda(), lens=lens, cache=cache).cpu().numpy()
        for b in range(B):
            got[b].append(s[b, :lens[b]])
            pos[b] += lens[b]
    for b in range(B):
        g = np.concatenate(got[b])
        assert g.shape == want[b][0].shape
        assert np.abs(g - want[b][0]).max() < TOL_F16, b
        c = cache[b].cpu().numpy().reshape(4, 19, 128)
        assert np.abs(c - want[b][1]).max() < 2e-2 * max(1.0, np.abs(want[b][1]).max()), b
    # chunked with a cache == one shot, to the last bit: a row's arithmetic does not depend on its neighbours
    one = net(torch.from_numpy(xs[0]).cuda()[None])[0].cpu().numpy()
    assert np.array_equal(np.concatenate(got[0]), one)
    # without a cache every call starts from silence
    a = net(torch.from_numpy(xs[1][:30]).cuda()[None])[0].cpu().numpy()
    b2 = net(torch.from_numpy(xs[1][:30]).cuda()[None], cache=net.new_cache(1, "cuda"))[0].cpu().numpy()
    assert np.array_equal(a, b2)


def test_argument_checks(pkg):
    sd = vo.make_state_dict(seed=14)
    net = make_net(pkg, sd)
    with pytest.raises(ValueError):
        net(torch.zeros((1, 5, 399), device="cuda"))
    with pytest.raises(ValueError):
        net(torch.zeros((2, 5, 400), device="cuda"), cache=net.new_cache(1, "cuda"))
    assert net(torch.zeros((0, 5, 400), device="cuda")).shape == (0, 5, 248)
    bad = dict(sd)
    del bad["encoder.fsmn.2.affine.linear.bias"]
    from toolbox_for_asr_and_tts_amd import _lib
    cfg = pkg.default_config()
    ctx = _lib.Context(cfg)
    with pytest.raises(pkg.AsrfeError, match="missing tensor encoder.fsmn.2.affine.linear.bias"):
        pkg.FsmnVadNet(ctx).load_state_dict(bad)
    with pytest.raises(ValueError, match="rorder"):
        pkg.FsmnVadNet(ctx, rorder=2)
    # a context without the weights refuses the forward
    with pytest.raises((pkg.AsrfeError, ValueError)):
        pkg.FsmnVadNet(ctx)(torch.zeros((1, 5, 400), device="cuda"))


def test_frame_level_kernel(pkg):
    from toolbox_for_asr_and_tts_amd import vad
    net = make_net(pkg, vo.make_state_dict(seed=15))
    clips = [vu.bursts(16000, [(4000, 9000)], seed=1), vu.bursts(5000, [], seed=2), np.zeros(399, np.int16),
             np.zeros(400, np.int16)]
    n = max(len(c) for c in clips)
    pcm = np.zeros((len(clips), n), dtype=np.int16)
    for i, c in enumerate(clips):
        pcm[i, :len(c)] = c
    lens = [len(c) for c in clips]
    got = vad.frame_decibel(net.ctx, torch.from_numpy(pcm).cuda(), lens).cpu().numpy()
    assert got.shape == (4, fo.num_frames(n))
    for i, c in enumerate(clips):
        want = vo.frame_decibel(c)
        assert np.abs(got[i, :len(want)] - want).max() < 1e-3 if len(want) else True
        assert np.all(got[i, len(want):] == 0.0)
    assert abs(got[3, 0] - (-60.0)) < 1e-4               # an all-zero frame: 10 log10(1e-6)
    # float PCM in [-1, 1] gives the same levels
    gotf = vad.frame_decibel(net.ctx, torch.from_numpy(pcm.astype(np.float32) / 32768.0).cuda(), lens).cpu().numpy()
    assert np.abs(gotf - got).max() < 1e-4


def test_streaming_frontend_rows_equal_the_oracle(pkg):
    from toolbox_for_asr_and_tts_amd import vad
    model = pkg.FsmnVadModel(vu.vad_config(), vu.cmvn(), vu.energy_state_dict())
    pcm = np.asarray(fo.synth_pcm("bursts", 16000 * 2 + 77, 9), dtype=np.int16)
    shift, scale = vu.cmvn()
    want = (vo.lfr_offline(fo.kaldi_fbank(pcm)) + shift) * scale
    for chunk in (6400, 1000, 250):
        fe = vad._OnlineFrontend(model.frontend, model.device)
        ofe = vo.OnlineFrontend()
        rows = []
        pieces = [pcm[i:i + chunk] for i in range(0, len(pcm), chunk)]
        for i, piece in enumerate(pieces):
            f, d = fe.push(piece, i == len(pieces) - 1)
            of, od = ofe.push(piece, i == len(pieces) - 1)
            assert f.shape[0] == of.shape[0] == d.shape[0], (chunk, i)       # the same rows become available per call
            if len(od):
                assert np.abs(d - od).max() < 1e-3
            rows.append(f.cpu().numpy())
        got = np.concatenate(rows)
        assert got.shape == want.shape
        assert np.abs(got - want).max() < 1e-3, chunk


def test_generate_per_chunk_as_the_reference_calls_it(pkg, tmp_path):
    vu.write_vad_dir(str(tmp_path / "fsmn_vad"))
    vad_model = pkg.AutoModel(model=str(tmp_path / "fsmn_vad"), device="cuda:0", disable_update=True, disable_pbar=True)
    pcm = vu.bursts(16000 * 8, [(16000, 48000), (80000, 100000)], seed=3)
    audio = pcm.astype(np.float32) / 32768.0                      # the session's float chunks
    # one shot (chunk_size default: not a streaming input, final)
    whole = vad_model.generate(input=audio)
    assert whole[0]["key"] == "rand_key_0"
    segs = whole[0]["value"]
    assert len(segs) == 2
    assert abs(segs[0][0] - 740) <= 40 and abs(segs[0][1] - 3160) <= 40
    assert abs(segs[1][0] - 4740) <= 40 and abs(segs[1][1] - 6410) <= 40
    # the oracle's trace through the same detector gives the same segments
    sil, db = vu.oracle_sil_trace(pcm, vu.energy_state_dict())
    assert pkg.VadDetector().step(sil, db, True, False) == segs
    # per 400 ms chunk with a cache, is_final=False: [beg, -1] when a segment opens, [-1, end] when it closes
    cache = {}
    events = []
    n_speech_chunks = 0
    for i in range(0, len(audio), 6400):
        chunk = audio[i:i + 6400]
        res = vad_model.generate(input=chunk, cache=cache, is_final=False, chunk_size=int(len(chunk) / 16000 * 1000))
        assert isinstance(res, list) and isinstance(res[0], dict)
        value = res[0].get("value", [])
        n_speech_chunks += len(value) > 0                         # what the reference reads (voice_interface.py:1596-1598)
        events += value
    assert events == [[segs[0][0], -1], [-1, segs[0][1]], [segs[1][0], -1], [-1, segs[1][1]]]
    assert n_speech_chunks == 4
    # a final call on the same cache closes the stream and clears the cache
    assert vad_model.generate(input=np.zeros(0, np.float32), cache=cache, is_final=True, chunk_size=400)[0]["value"] == []
    assert "_stream" not in cache
    # chunks that are not a multiple of the chunk size: the remainder waits for the next call
    cache = {}
    ev2 = []
    for i in range(0, len(audio), 5000):
        last = i + 5000 >= len(audio)
        ev2 += vad_model.generate(input=audio[i:i + 5000], cache=cache, is_final=last, chunk_size=400)[0]["value"]
    assert ev2 == events
    # a batch of clips without caches: one result each
    two = vad_model.generate(input=[pcm, pcm[:40000]], key=["a", "b"])
    assert [r["key"] for r in two] == ["a", "b"] and two[0]["value"] == segs and len(two[1]["value"]) == 1
    with pytest.raises(ValueError, match="one input"):
        vad_model.generate(input=[pcm, pcm], cache={})


def test_streaming_session_with_the_vad_model(pkg, tmp_path):
    """The session asks the model once per 400 ms chunk (voice_interface.py:1585-1590) and keeps deciding by energy."""
    vu.write_vad_dir(str(tmp_path / "v"))
    vad_model = pkg.AutoModel(model=str(tmp_path / "v"), device="cuda:0")
    ses = pkg.StreamingPcmSession(now=0.0, vad_model=vad_model)
    # 1 s quiet, 2 s of noise at 0.1 (the energy rule needs mean 0.03 / max 0.17: silent for it), 2 s quiet
    rng = np.random.default_rng(0)
    audio = (rng.standard_normal(16000 * 5) * 1e-4).astype(np.float32)
    audio[16000:48000] = (rng.standard_normal(32000) * 0.03).astype(np.float32)
    flags = []
    for i in range(0, len(audio), 6400):
        st = ses.process_chunk(audio[i:i + 6400], now=0.4 * (i // 6400 + 1))
        flags.append((st["is_speech"], ses.is_speech_vad))
    assert not any(f[0] for f in flags)                        # the energy rule never fires on this level
    assert sum(f[1] for f in flags) == 2                       # the model reports the segment's opening and its closing
    assert "_stream" in ses.vad_cache
    ses.reset(now=10.0)
    assert ses.vad_cache == {}


def test_many_streams_at_once_equal_single_streams(pkg):
    """64 streams x 1000 frames in one call (250 row tiles per product, every CU busy): a stream's scores and cache do not
    depend on its position in the batch or on its neighbours, to the last bit; and they match the oracle."""
    sd = vo.make_state_dict(seed=21)
    net = make_net(pkg, sd)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(64, 1000, 400, generator=g)
    lens = [1000 - 13 * (i % 7) for i in range(64)]
    cache = net.new_cache(64, "cuda")
    s = net(x.cuda(), lens=lens, cache=cache).cpu()
    for b in (0, 31, 63):
        c1 = net.new_cache(1, "cuda")
        s1 = net(x[b:b + 1, :lens[b]].contiguous().cuda(), cache=c1).cpu()
        assert torch.equal(s[b, :lens[b]], s1[0])
        assert torch.equal(cache[b].cpu(), c1[0].cpu())
    want, _ = vo.fsmn_forward(sd, x[31, :lens[31]].numpy())
    assert np.abs(s[31, :lens[31]].numpy() - want).max() < TOL_F16
