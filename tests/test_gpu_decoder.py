"""GPU parity of the step behind the encoder (CIF predictor + Paraformer SAN-M decoder + arg-max), called through the
C ABI, against the CPU oracle (oracle/paraformer_decoder_oracle.py, fp64) and the committed fixture.  Bars: alphas to
1e-5, token counts exact, token embeddings to 1e-4, logits to 3e-2 (16-bit MFMA operands through up to 17 blocks and
the vocabulary projection), token ids exact wherever the oracle's best logit leads the runner-up by more than twice
that."""
import os

import numpy as np
import pytest
import torch

from oracle import paraformer_decoder_oracle as po

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
ALPHA_TOL, EMB_TOL, LOGIT_TOL = 1e-5, 1e-4, 3e-2


@pytest.fixture(scope="module")
def pkg():
    import toolbox_for_asr_and_tts_amd as p

    p.load()
    return p


# launch plans of the matrix products: the library's default (calls of up to 2048 rows take the latency plan's product)
# and the throughput kernel only
PLANS = [None, 0]


def make_head(pkg, cfg, sd, small_rows=None):
    from toolbox_for_asr_and_tts_amd.decoder import ParaformerHead
    lib_cfg = pkg.default_config()
    if small_rows is not None:
        lib_cfg.enc_small_batch_rows = small_rows
    head = ParaformerHead(pkg.Context(lib_cfg), num_blocks=cfg.num_blocks, linear_units=cfg.linear_units, vocab_size=cfg.vocab_size)
    head.load_state_dict(sd)
    return head


def check_against_oracle(head, cfg, sd, mem, lens, n_max=None):
    sd64 = {k: v.double() for k, v in sd.items()}
    want_ids, want_logits, want_tn, want_alphas = po.recognize(mem.double(), lens, sd64, cfg)
    want_emb, _, _, _ = po.predictor_forward(mem.double(), lens, sd64, cfg)
    r = head.forward(mem.cuda(), lens, n_max=n_max, return_logits=True, return_alphas=True)
    torch.cuda.synchronize()
    assert float((r["alphas"].cpu().double() - want_alphas).abs().max()) <= ALPHA_TOL
    assert r["token_num"].cpu().tolist() == want_tn.tolist()
    worst = 0.0
    for k in range(mem.shape[0]):
        n = int(want_tn[k])
        if n == 0:
            assert torch.all(r["token_ids"][k] == -1)
            continue
        assert float((r["embeds"][k, :n].cpu().double() - want_emb[k, :n]).abs().max()) <= EMB_TOL
        assert float(r["embeds"][k, n:].abs().max()) == 0.0 if n < r["embeds"].shape[1] else True
        got = r["logits"][k, :n].cpu().double()
        err = float((got - want_logits[k, :n]).abs().max())
        worst = max(worst, err)
        assert err <= LOGIT_TOL, (k, err)
        top2 = want_logits[k, :n].topk(2, dim=-1).values
        sure = (top2[:, 0] - top2[:, 1]) > 2 * LOGIT_TOL
        ids = r["token_ids"][k, :n].cpu()
        assert torch.equal(ids[sure], want_logits[k, :n].argmax(-1)[sure].to(torch.int32))
        assert torch.equal(ids, got.argmax(-1).to(torch.int32))          # the arg-max kernel agrees with its own logits
        assert torch.all(r["token_ids"][k, n:] == -1)
    return worst


@pytest.mark.parametrize("plan", PLANS)
def test_decoder_tiny_fixture(pkg, plan):
    g = np.load(os.path.join(GOLDEN, "decoder_tiny.npz"))
    cfg = po.DecoderConfig(num_blocks=int(g["num_blocks"]), vocab_size=int(g["vocab_size"]))
    sd = po.make_state_dict(cfg, seed=int(g["seed"]))
    head = make_head(pkg, cfg, sd, plan)
    mem = torch.from_numpy(g["memory"])
    r = head.forward(mem.cuda(), torch.from_numpy(g["lens"]), return_logits=True, return_alphas=True)
    torch.cuda.synchronize()
    assert np.max(np.abs(r["alphas"].cpu().numpy() - g["alphas"])) <= ALPHA_TOL
    assert r["token_num"].cpu().tolist() == g["token_num"].tolist()
    for k in range(2):
        n = int(g["token_num"][k])
        assert np.max(np.abs(r["embeds"][k, :n].cpu().numpy() - g["embeds"][k, :n])) <= EMB_TOL
        assert np.max(np.abs(r["logits"][k, :n].cpu().numpy() - g["logits"][k, :n])) <= LOGIT_TOL
    assert r["logits"].shape[2] == 300 and head.logits_stride == 512


@pytest.mark.parametrize("plan", PLANS)
def test_predictor_and_decoder_ragged_lengths(pkg, plan):
    """Lengths around the edges: an empty utterance (no tokens), a single frame, lengths that end a 64-key tile."""
    cfg = po.DecoderConfig(num_blocks=2, vocab_size=515)
    sd = po.make_state_dict(cfg, seed=11)
    head = make_head(pkg, cfg, sd, plan)
    g = torch.Generator().manual_seed(12)
    mem = torch.randn(5, 130, 512, generator=g)
    lens = torch.tensor([130, 64, 1, 0, 65], dtype=torch.int32)
    check_against_oracle(head, cfg, sd, mem, lens)


@pytest.mark.parametrize("plan", PLANS)
def test_full_size_decoder(pkg, plan):
    """The paraformer-large head: 16 blocks + closing block, linear_units 2048, vocabulary 8404; B = 2, T' = 100."""
    cfg = po.DecoderConfig()
    sd = po.make_state_dict(cfg, seed=13)
    head = make_head(pkg, cfg, sd, plan)
    g = torch.Generator().manual_seed(14)
    mem = torch.randn(2, 100, 512, generator=g)
    lens = torch.tensor([100, 73], dtype=torch.int32)
    worst = check_against_oracle(head, cfg, sd, mem, lens)
    print(f"full-size decoder: worst logit error {worst:.3e}")
    # n_max below the token count truncates: the first n_max tokens are the same tokens
    full = head.forward(mem.cuda(), lens)
    n_cut = 12
    cut = head.forward(mem.cuda(), lens, n_max=n_cut, return_logits=True)
    torch.cuda.synchronize()
    assert cut["token_num"].cpu().tolist() == [min(int(t), n_cut) for t in full["token_num"].cpu().tolist()]
    assert torch.equal(cut["embeds"], full["embeds"][:, :n_cut])


def test_rows_beyond_one_round_of_half_height_tiles(pkg):
    """More than 16384 token rows: the products with 512 output columns then take full-height tiles, whose after-the-fact
    LayerNorm gathers its row statistics in front of the epilogue instead of behind the main loop.  72 copies of four
    utterances (288 x 64 = 18432 rows): every copy gives the logits of the four-utterance call (the same sums in the same
    order whatever the tile height), and those meet the oracle."""
    cfg = po.DecoderConfig(num_blocks=2, vocab_size=515)
    sd = po.make_state_dict(cfg, seed=41)
    head = make_head(pkg, cfg, sd, 0)
    g = torch.Generator().manual_seed(42)
    mem = torch.randn(4, 150, 512, generator=g)
    lens = torch.tensor([150, 97, 1, 128], dtype=torch.int32)
    check_against_oracle(head, cfg, sd, mem, lens)
    small = head.forward(mem.cuda(), lens, n_max=64, return_logits=True)  # (n_max truncates: the first 64 tokens)
    big = head.forward(mem.repeat(72, 1, 1).cuda(), lens.repeat(72), n_max=64, return_logits=True)
    torch.cuda.synchronize()
    assert big["token_ids"].shape == (288, 64) and 288 * 64 > 16384
    for c in (0, 35, 71):
        assert torch.equal(big["token_num"][4 * c:4 * c + 4], small["token_num"])
        assert torch.equal(big["token_ids"][4 * c:4 * c + 4], small["token_ids"])
        for k in range(4):
            n = int(small["token_num"][k])
            if n:
                assert float((big["logits"][4 * c + k, :n] - small["logits"][k, :n]).abs().max()) <= 1e-5


def test_head_and_vocoder_side_by_side_repeat_bit_for_bit(pkg):
    """The decoder head at the bench's shape (64 x 500 frames, 16 blocks, vocabulary 8404) on one stream while the vocoder
    (128 x 94 frames) runs on another: every repetition of either gives the bits of its run alone.  The head's products
    are the general GEMM on half-height tiles with operands the launch before has just written, the case in which a
    fragment read left in flight across the GEMM's phase barrier once lost against the refill of its half-tile; a
    second stream on the same chip is what made that visible in the encoder."""
    from oracle import hifigan_oracle as ho
    from toolbox_for_asr_and_tts_amd.vocoder import HiFiGANGenerator
    cfg = po.DecoderConfig()
    head = make_head(pkg, cfg, po.make_state_dict(cfg, seed=51), 0)
    g = torch.Generator().manual_seed(52)
    mem = (torch.randn(64, 500, 512, generator=g) * 0.7).cuda()
    lens = torch.full((64,), 500, dtype=torch.int32)
    vcfg = ho.HifiganConfig()
    gen = HiFiGANGenerator(n_mels=vcfg.n_mels, initial_channel=vcfg.initial_channel, upsample_rates=vcfg.upsample_rates,
                           upsample_kernels=vcfg.upsample_kernels, resblock_kernels=vcfg.resblock_kernels,
                           resblock_dilations=vcfg.resblock_dilations, lrelu_slope=vcfg.lrelu_slope, final_slope=vcfg.final_slope)
    gen.load_state_dict(ho.make_state_dict(vcfg, seed=53))
    mel = torch.randn(128, 80, 94, generator=g).cuda()

    def head_run():
        r = head.forward(mem, lens, n_max=256, return_alphas=True)
        return [r["token_ids"].clone(), r["token_num"].clone(), r["embeds"].clone(), r["alphas"].clone()]

    want_head = head_run()
    want_wav = gen.forward(mel).clone()
    torch.cuda.synchronize()
    assert int(want_head[1].sum()) > 0 and bool(torch.isfinite(want_wav).all())
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for it in range(6):
        sa.wait_stream(torch.cuda.current_stream())
        sb.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(sa):
            got_head = head_run()
        with torch.cuda.stream(sb):
            got_wav = gen.forward(mel).clone()
        torch.cuda.synchronize()
        for a, b in zip(got_head, want_head):
            assert torch.equal(a, b), it
        assert torch.equal(got_wav, want_wav), it


@pytest.mark.parametrize("vocab", [515, 8404])
def test_token_ids_without_logits_equal_those_with(pkg, vocab):
    """A caller that does not ask for the logits gets its ids from the vocabulary projection's own per-strip maxima
    (throughput plan; nothing of the projection is stored): the same ids as with the logits, for a vocabulary that ends
    inside a strip, with an empty utterance in the batch; and with all logits of a row equal the lowest index wins."""
    cfg = po.DecoderConfig(num_blocks=1, vocab_size=vocab)
    sd = po.make_state_dict(cfg, seed=31)
    g = torch.Generator().manual_seed(32)
    mem = torch.randn(4, 90, 512, generator=g)
    lens = torch.tensor([90, 0, 64, 17], dtype=torch.int32)
    head = make_head(pkg, cfg, sd, 0)
    a = head.forward(mem.cuda(), lens, return_logits=True)
    b = head.forward(mem.cuda(), lens)
    torch.cuda.synchronize()
    assert "logits" not in b
    assert torch.equal(a["token_ids"], b["token_ids"])
    assert torch.equal(a["token_num"], b["token_num"]) and int(a["token_num"].sum()) > 0
    if vocab == 8404:  # enough token rows for the projection to take its full-height tiles (more than 256 half-height ones)
        mem_l = torch.randn(12, 240, 512, generator=g)
        lens_l = torch.tensor([240, 200, 240, 90, 240, 1, 240, 177, 240, 240, 64, 240], dtype=torch.int32)
        al = head.forward(mem_l.cuda(), lens_l, return_logits=True)
        bl = head.forward(mem_l.cuda(), lens_l)
        torch.cuda.synchronize()
        rows = al["token_ids"].shape[0] * al["token_ids"].shape[1]
        assert (rows + 127) // 128 * 33 > 256, rows
        assert torch.equal(al["token_ids"], bl["token_ids"])
        assert torch.equal(al["token_ids"][al["token_ids"] >= 0].long(),
                           al["logits"].argmax(-1)[al["token_ids"] >= 0])
    # ties: a constant output layer; the last vocabulary entry alone is raised in a second run
    sd2 = dict(sd)
    sd2["decoder.output_layer.weight"] = torch.zeros_like(sd["decoder.output_layer.weight"])
    sd2["decoder.output_layer.bias"] = torch.full_like(sd["decoder.output_layer.bias"], 0.25)
    head2 = make_head(pkg, cfg, sd2, 0)
    for with_logits in (True, False):
        r = head2.forward(mem.cuda(), lens, return_logits=with_logits)
        ids = r["token_ids"]
        for k in range(4):
            n = int(r["token_num"][k])
            assert torch.all(ids[k, :n] == 0) and torch.all(ids[k, n:] == -1)
    sd2["decoder.output_layer.bias"] = sd2["decoder.output_layer.bias"].clone()
    sd2["decoder.output_layer.bias"][vocab - 1] = 0.5
    head3 = make_head(pkg, cfg, sd2, 0)
    for with_logits in (True, False):
        r = head3.forward(mem.cuda(), lens, return_logits=with_logits)
        for k in range(4):
            n = int(r["token_num"][k])
            assert torch.all(r["token_ids"][k, :n] == vocab - 1)


def test_engine_to_token_ids(pkg):
    """PCM -> frontend -> encoder -> predictor -> decoder on the device; the head's result equals the oracle head run
    on the device encoder's own output (the oracle encoder's output differs by the encoder tolerance, which moves
    integrate-and-fire decisions: the two halves are held to their own bars)."""
    from oracle import fbank_oracle as fo
    from oracle import sanm_oracle as so
    from toolbox_for_asr_and_tts_amd.decoder import ParaformerHead
    n = 48000
    clips = np.stack([fo.synth_pcm("bursts", n, 70), fo.synth_pcm("gauss", n, 71)])
    lens = [48000, 30000]
    cfg_e = pkg.default_config()
    cfg_e.enc_num_blocks = 2
    eng = pkg.AsrEngine(batch=2, max_samples=n, cfg=cfg_e, state_dict=so.make_state_dict(so.SanmConfig(num_blocks=2), seed=2))
    cfg = po.DecoderConfig(num_blocks=2, vocab_size=300)
    sd = po.make_state_dict(cfg, seed=15)
    head = ParaformerHead(eng.ctx, num_blocks=2, vocab_size=300)     # shares the engine's context
    head.load_state_dict(sd)
    out, olens = eng.run(torch.from_numpy(clips).cuda(), torch.tensor(lens, dtype=torch.int32, device="cuda"))
    ids = head.recognize(out, olens)
    torch.cuda.synchronize()
    want_ids, _, want_tn, _ = po.recognize(out.cpu().double(), olens.cpu(), {k: v.double() for k, v in sd.items()}, cfg)
    assert [len(i) for i in ids] == [len(i) for i in want_ids]
    # every token whose best logit leads the runner-up by more than twice the logit tolerance in the oracle must come
    # out the same (random-weight logits are nearly flat, so some tokens are near-ties: those may differ, and say nothing)
    _, want_logits, _, _ = po.recognize(out.cpu().double(), olens.cpu(), {k: v.double() for k, v in sd.items()}, cfg)
    r = head.forward(out, olens)
    n_sure = 0
    for k in range(2):
        n = int(want_tn[k])
        top2 = want_logits[k, :n].topk(2, dim=-1).values
        sure = (top2[:, 0] - top2[:, 1]) > 2 * LOGIT_TOL
        n_sure += int(sure.sum())
        assert torch.equal(r["token_ids"][k, :n].cpu()[sure], want_logits[k, :n].argmax(-1)[sure].to(torch.int32))
    assert n_sure >= 10
    # with a token list the ids become text
    head.token_list = [f"t{i}" if i % 3 else chr(0x4E00 + i) for i in range(300)]
    texts = head.recognize(out, olens)
    assert len(texts) == 2 and all(isinstance(t, str) and t for t in texts)


def test_decoder_refusals(pkg):
    from toolbox_for_asr_and_tts_amd.decoder import ParaformerHead
    cfg = po.DecoderConfig(num_blocks=1, vocab_size=300)
    sd = po.make_state_dict(cfg, seed=16)
    head = ParaformerHead(pkg.Context(), num_blocks=1, vocab_size=300)
    with pytest.raises(ValueError):
        head.workspace_bytes(1, 10, 11)                      # nothing loaded yet
    bad = dict(sd)
    del bad["decoder.decoders3.0.norm1.weight"]
    with pytest.raises(pkg.AsrfeError, match="missing tensor decoder.decoders3.0.norm1.weight"):
        head.load_state_dict(bad)
    big = dict(sd)
    big["decoder.decoders.0.feed_forward.w_1.weight"] = sd["decoder.decoders.0.feed_forward.w_1.weight"] * 1e5
    with pytest.raises(pkg.AsrfeError, match="feed-forward hidden layer.*fp16 range"):
        ParaformerHead(pkg.Context(), num_blocks=1, vocab_size=300).load_state_dict(big)
    head.load_state_dict(sd)
    with pytest.raises(pkg.AsrfeError, match="already loaded"):
        head.load_state_dict(sd)
    with pytest.raises(ValueError):
        head.forward(torch.zeros(1, 10, 512), [10])          # host tensor


def test_http_route_returns_the_heads_text(pkg):
    """POST /api/v1/voice/asr with a ParaformerHead as the default decoder: success = true and the text is what the
    head recognises on the engine's output for the clip."""
    import base64
    import importlib
    import io
    import wave
    from fastapi.testclient import TestClient
    from oracle import fbank_oracle as fo
    from oracle import sanm_oracle as so
    from toolbox_for_asr_and_tts_amd.decoder import ParaformerHead
    svc = importlib.import_module("toolbox_for_asr_and_tts_amd.service")
    pcm = fo.synth_pcm("bursts", 16000 * 2, 21)
    buf = io.BytesIO()
    with wave.open(buf, "wb") as wf:
        wf.setnchannels(1)
        wf.setsampwidth(2)
        wf.setframerate(16000)
        wf.writeframes(pcm.tobytes())
    cfg_e = pkg.default_config()
    cfg_e.enc_num_blocks = 2
    eng = pkg.AsrEngine(batch=1, max_samples=16000 * 5, cfg=cfg_e, state_dict=so.make_state_dict(so.SanmConfig(num_blocks=2), seed=8))
    cfg = po.DecoderConfig(num_blocks=2, vocab_size=300)
    head = ParaformerHead(eng.ctx, num_blocks=2, vocab_size=300, token_list=[chr(0x4E00 + i) for i in range(300)])
    head.load_state_dict(po.make_state_dict(cfg, seed=17))
    out, olens = eng.run(torch.from_numpy(pcm[None]).cuda(), torch.tensor([pcm.size], dtype=torch.int32, device="cuda"))
    want = head.recognize(out, olens)[0]
    assert isinstance(want, str) and len(want) > 0
    client = TestClient(svc.create_app(eng, head=head))
    r = client.post("/api/v1/voice/asr", json={"audio_data": base64.b64encode(buf.getvalue()).decode(), "diagnosis_session_id": "t"})
    assert r.status_code == 200 and r.json() == {"text": want, "success": True, "message": None, "sample_id": None}


def _raw_calls(pkg, head, mem, lens, n_max, ws_pred, ws_dec, embeds_override=None):
    """asrfe_cif_predictor and asrfe_decoder_forward through the C ABI with a workspace each."""
    import ctypes as C
    ctx = head.ctx
    b, tp, _ = mem.shape
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    embeds = torch.empty((b, n_max, 512), dtype=torch.float32, device="cuda")
    token_num = torch.zeros((b,), dtype=torch.int32, device="cuda")
    ids = torch.full((b, n_max), -7, dtype=torch.int32, device="cuda")
    alphas = torch.empty((b, tp + 1), dtype=torch.float32, device="cuda")
    ctx.check(ctx.lib.asrfe_cif_predictor(ctx.handle, mem.data_ptr(), lens.data_ptr(), b, tp, n_max, embeds.data_ptr(),
                                          token_num.data_ptr(), alphas.data_ptr(), ws_pred.data_ptr(), ws_pred.numel(), st))
    if embeds_override is not None:
        embeds = embeds_override(embeds)
    ctx.check(ctx.lib.asrfe_decoder_forward(ctx.handle, mem.data_ptr(), lens.data_ptr(), b, tp, embeds.data_ptr(),
                                            token_num.data_ptr(), n_max, ids.data_ptr(), None, ws_dec.data_ptr(),
                                            ws_dec.numel(), st))
    torch.cuda.synchronize()
    return ids, token_num, alphas


@pytest.mark.parametrize("plan", PLANS)
def test_decoder_forward_needs_nothing_from_the_predictors_workspace(pkg, plan):
    """asrfe_decoder_forward builds its own 16-bit copy of the encoder frames: run on a workspace of its own that is full
    of NaN patterns it returns the ids of the call that shares the predictor's workspace."""
    cfg = po.DecoderConfig(num_blocks=2, vocab_size=300)
    sd = po.make_state_dict(cfg, seed=41)
    head = make_head(pkg, cfg, sd, plan)
    g = torch.Generator().manual_seed(42)
    mem = torch.randn(3, 70, 512, generator=g).cuda()
    lens = torch.tensor([70, 33, 64], dtype=torch.int32, device="cuda")
    n_max = 71
    nbytes = head.workspace_bytes(3, 70, n_max)
    ws_a = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    ws_b = torch.full((nbytes,), 0xFF, dtype=torch.uint8, device="cuda")   # 0xFFFF...: NaN as fp16, bf16 and f32
    same, tn, _ = _raw_calls(pkg, head, mem, lens, n_max, ws_a, ws_a)
    other, tn2, _ = _raw_calls(pkg, head, mem, lens, n_max, ws_a, ws_b)
    assert torch.equal(tn, tn2) and int(tn.sum()) > 10
    assert torch.equal(same, other)
    assert int((same >= 0).sum()) == int(tn.sum())


def test_a_row_without_a_maximum_reports_a_vocabulary_index(pkg):
    """Logits at the bottom of the float range (an output bias of -inf) have no entry above the arg-max's starting
    value: the id of such a row is 0 (blank), never a number outside the vocabulary, on both launch plans, with and
    without stored logits.  (NaN embeddings do not reach
    the logits: the closing block has no residual path and its ReLU returns 0 for a NaN.)"""
    cfg = po.DecoderConfig(num_blocks=1, vocab_size=300)
    sd = po.make_state_dict(cfg, seed=43)
    sd["decoder.output_layer.bias"] = torch.full_like(sd["decoder.output_layer.bias"], float("-inf"))
    g = torch.Generator().manual_seed(44)
    mem = torch.randn(2, 60, 512, generator=g).cuda()
    lens = torch.tensor([60, 41], dtype=torch.int32)
    for plan in PLANS:
        head = make_head(pkg, cfg, sd, plan)
        for with_logits in (False, True):
            r = head.forward(mem, lens, return_logits=with_logits)
            torch.cuda.synchronize()
            for k in range(2):
                n = int(r["token_num"][k])
                assert n > 3 and torch.all(r["token_ids"][k, :n] == 0) and torch.all(r["token_ids"][k, n:] == -1)
            if with_logits:
                assert float(r["logits"][0, 0].max()) < -3e38     # (the loader keeps the bias finite: -FLT_MAX)
            assert head.clean_ids(r["token_ids"][0].cpu().tolist(), 300) == []


def test_fire_frames_equal_the_oracles(pkg):
    """asrfe_cif_fire_frames on the predictor's own alphas: the frames equal those of the float32 recurrence on the same
    alphas (exact: the same additions in the same order), one fire per token, rising, inside the utterance (the tail
    weight may fire at position lens[b])."""
    cfg = po.DecoderConfig(num_blocks=1, vocab_size=300)
    sd = po.make_state_dict(cfg, seed=45)
    head = make_head(pkg, cfg, sd)
    g = torch.Generator().manual_seed(46)
    mem = torch.randn(4, 130, 512, generator=g)
    lens = torch.tensor([130, 64, 0, 97], dtype=torch.int32)
    r = head.forward(mem.cuda(), lens, return_alphas=True, return_fire_frames=True)
    torch.cuda.synchronize()
    fires = r["fire_frames"].cpu()
    want = po.cif_fire_frames(r["alphas"].cpu(), cfg.threshold, fires.shape[1])
    assert torch.equal(fires, want)
    for k in range(4):
        n = int(r["token_num"][k])
        f = fires[k][fires[k] >= 0].tolist()
        assert len(f) >= n and f == sorted(set(f)) and all(0 <= v <= int(lens[k]) for v in f)
    # a smaller n_max keeps the first fires
    cut = head.forward(mem.cuda(), lens, n_max=5, return_fire_frames=True)["fire_frames"].cpu()
    assert torch.equal(cut, fires[:, :5])
