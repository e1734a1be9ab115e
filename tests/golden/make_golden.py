"""Generate the committed golden fixtures under tests/golden/.

Run from the repo root:  python tests/golden/make_golden.py

The reference holds no golden vectors for this path and its arithmetic lives in
packages that are not installed here, so the fixtures are produced by the CPU
oracle (oracle/fbank_oracle.py, oracle/sanm_oracle.py) and, for fbank, checked
while generating against an independent Kaldi-fbank implementation that IS
importable in the build container: ``transformers.audio_utils.spectrogram`` in the
configuration its own feature extractors use to stand in for
``torchaudio.compliance.kaldi.fbank`` (pre-emphasis 0.97, DC removal, Kaldi mel
scale, triangles in mel space, FLT_EPSILON floor, natural log) with a symmetric
Hamming window.  Only the fixtures travel to the GPU box; transformers is not needed
there.

Files written:
  fbank_<kind>_s<seed>_<n>.npz : pcm int16, logmel f32 (T, 80), feats f32 (T', 560), shift/scale f32
  encoder_tiny.npz             : 2-layer SAN-M encoder, B=2, T'=40 ragged: feats, lens, out (fp64 oracle -> f32)
  encoder_probe_50l.npz        : 50-layer encoder, B=2, T'=96 ragged: out slices + checksums
  encoder_full_50l_t500.npz    : 50-layer encoder at the full frame count, B=2, T'=500 with lens [500, 333]
                                 (fp64 oracle): selected rows, per-clip column sums, mean |out|
  bench_clips_50l.npz          : clips 0, 31 and 63 of the benchmark batch (synth.bench_pcm(64, 480000, 1234), 30 s
                                 each) through the whole oracle chain, fbank -> LFR -> CMVN (identity) -> 50 blocks in
                                 fp64 with the benchmark's weights: selected feature rows and encoder rows, column sums

  decoder_tiny.npz             : CIF predictor + 2-block decoder (vocabulary 300), B=2, T'=48 ragged: memory, alphas,
                                 token counts, embeddings, logits, arg-max ids (fp64 oracle)

``python tests/golden/make_golden.py witness`` writes nothing: it checks the oracles whose modules have a second,
independent implementation in the build image against it (second_witnesses below; tests/test_second_witness_cpu.py runs
the same functions).

``python tests/golden/make_golden.py full`` writes only the two full-size encoder fixtures (several minutes of CPU
time), ``... decoder`` only the decoder fixture, ``... vocoder`` only the vocoder fixture
(vocoder_tiny.npz: HiFi-GAN generator with 128 channels and two stages, B=2, 16 mel frames: mel, waveform).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import fbank_oracle as fo  # noqa: E402
from oracle import sanm_oracle as so  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

FBANK_CASES = [
    ("uniform", 0, 16000),
    ("gauss", 1, 16000),
    ("chirp", 2, 48000),
    ("bursts", 3, 48000),
    ("bench", 4, 16000),
    ("gauss", 5, 400),  # exactly one frame
    ("gauss", 6, 1367),  # ragged: 7 frames -> 2 LFR frames with tail replication
]


def cross_check_fbank(pcm: np.ndarray, ours: np.ndarray) -> float:
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function

    mel_filters = mel_filter_bank(
        num_frequency_bins=257,
        num_mel_filters=80,
        min_frequency=20,
        max_frequency=8000,
        sampling_rate=16000,
        norm=None,
        mel_scale="kaldi",
        triangularize_in_mel_space=True,
    )
    window = window_function(400, "hamming", periodic=False)
    theirs = spectrogram(
        pcm.astype(np.float64),
        window,
        frame_length=400,
        hop_length=160,
        fft_length=512,
        power=2.0,
        center=False,
        preemphasis=0.97,
        mel_filters=mel_filters,
        log_mel="log",
        mel_floor=1.192092955078125e-07,
        remove_dc_offset=True,
        dtype=np.float64,
    ).T
    assert theirs.shape == ours.shape, (theirs.shape, ours.shape)
    return float(np.max(np.abs(theirs - ours)))


def synth_cmvn(seed: int):
    rng = np.random.default_rng(1000 + seed)
    shift = -rng.uniform(6.0, 14.0, size=560)
    scale = 1.0 / rng.uniform(1.5, 4.5, size=560)
    return shift, scale


FULL_ROWS = [0, 1, 2, 49, 50, 127, 128, 255, 256, 332, 333, 400, 498, 499]
BENCH_CLIPS = [0, 31, 63]


def full_size_goldens():
    """The two full-size encoder fixtures (50 blocks, T' = 500)."""
    cfg = so.SanmConfig(num_blocks=50)
    sd = so.make_state_dict(cfg, seed=0, dtype=torch.float64)
    feats = so.synth_feats(2, 500, 560, seed=13, dtype=torch.float64)
    lens = torch.tensor([500, 333], dtype=torch.int32)
    out, olens = so.encoder_forward(feats, lens, sd, cfg)
    o = out.numpy()
    rows1 = [r for r in FULL_ROWS if r < 333]
    np.savez_compressed(
        os.path.join(OUT, "encoder_full_50l_t500.npz"),
        lens=lens.numpy(),
        rows=np.array(FULL_ROWS, np.int32),
        rows_b0=o[0, FULL_ROWS].astype(np.float32),
        rows1=np.array(rows1, np.int32),
        rows_b1=o[1, rows1].astype(np.float32),
        col_sums=o.sum(axis=1).astype(np.float64),
        abs_mean=np.float64(np.abs(o).mean()),
        seed=np.int32(0),
        feats_seed=np.int32(13),
        num_blocks=np.int32(50),
    )
    print("encoder full 50l t500:", tuple(out.shape), "abs mean", float(np.abs(o).mean()))

    # the benchmark's own batch: three of its 64 clips through the whole chain
    sys.path.insert(0, ROOT)
    from toolbox_for_asr_and_tts_amd.synth import bench_pcm, encoder_state_dict

    pcm = bench_pcm(64, 480000, seed=1234)[BENCH_CLIPS]
    f, fl = fo.frontend(pcm, [480000] * len(BENCH_CLIPS))
    sd = {k: v.double() for k, v in encoder_state_dict(num_blocks=50, seed=0).items()}
    out, olens = so.encoder_forward(torch.from_numpy(f).double(), torch.from_numpy(fl), sd, cfg)
    o = out.numpy()
    np.savez_compressed(
        os.path.join(OUT, "bench_clips_50l.npz"),
        clips=np.array(BENCH_CLIPS, np.int32),
        pcm_seed=np.int32(1234),
        weights_seed=np.int32(0),
        rows=np.array(FULL_ROWS, np.int32),
        feat_rows=f[:, FULL_ROWS].astype(np.float32),
        feat_col_sums=f.sum(axis=1).astype(np.float64),
        out_rows=o[:, FULL_ROWS].astype(np.float32),
        out_col_sums=o.sum(axis=1).astype(np.float64),
        out_abs_mean=np.float64(np.abs(o).mean()),
        olens=olens.numpy(),
    )
    print("bench clips 50l:", tuple(out.shape), "abs mean", float(np.abs(o).mean()))


def decoder_goldens():
    """CIF predictor + decoder, tiny: 2 blocks, vocabulary 300, B = 2, T' = 48 ragged (fp64 oracle)."""
    from oracle import paraformer_decoder_oracle as po

    cfg = po.DecoderConfig(num_blocks=2, vocab_size=300)
    sd = {k: v.double() for k, v in po.make_state_dict(cfg, seed=7).items()}
    g = torch.Generator().manual_seed(8)
    mem = torch.randn(2, 48, 512, generator=g, dtype=torch.float32).double()
    lens = torch.tensor([48, 31], dtype=torch.int32)
    ids, logits, tn, alphas = po.recognize(mem, lens, sd, cfg)
    emb, _, _, _ = po.predictor_forward(mem, lens, sd, cfg)
    n = int(tn.max())
    idm = np.full((2, n), -1, np.int32)
    for k in range(2):
        idm[k, : int(tn[k])] = logits[k, : int(tn[k])].argmax(-1).numpy()
    np.savez_compressed(
        os.path.join(OUT, "decoder_tiny.npz"),
        memory=mem.numpy().astype(np.float32), lens=lens.numpy(), token_num=tn.numpy(), alphas=alphas.numpy(),
        embeds=emb.numpy().astype(np.float32), logits=logits.numpy().astype(np.float32), ids=idm,
        seed=np.int32(7), num_blocks=np.int32(2), vocab_size=np.int32(300),
    )
    print("decoder tiny: token_num", tn.tolist(), "logits", tuple(logits.shape))


def vocoder_goldens():
    """HiFi-GAN generator, tiny: 128 channels, two stages (4, 2), B = 2, 16 mel frames (fp64 oracle)."""
    from oracle import hifigan_oracle as ho

    cfg = ho.HifiganConfig(initial_channel=128, upsample_rates=(4, 2), upsample_kernels=(8, 4))
    sd = {k: v.double() for k, v in ho.make_state_dict(cfg, seed=9).items()}
    mel = torch.randn(2, 80, 16, generator=torch.Generator().manual_seed(10), dtype=torch.float32)
    wav = ho.generator_forward(mel.double(), sd, cfg)
    np.savez_compressed(os.path.join(OUT, "vocoder_tiny.npz"), mel=mel.numpy(), wav=wav.numpy().astype(np.float32),
                        seed=np.int32(9), initial_channel=np.int32(128), rates=np.array([4, 2], np.int32),
                        kernels=np.array([8, 4], np.int32))
    print("vocoder tiny:", tuple(wav.shape), "abs max", float(wav.abs().max()))


# ---- second witnesses: independent implementations of the same published modules, importable in the build image ----
def witness_hifigan(initial_channel=128, rates=(4, 2), kernels=(8, 4), frames=16, seed=9):
    """The HiFi-GAN oracle against ``transformers.SpeechT5HifiGan`` (its own port of the published generator: conv_pre,
    leaky_relu -> transposed convolution -> mean of the residual blocks per stage, leaky_relu 0.01, conv_post, tanh) with
    the oracle's seeded weights, in fp64.  -> max |oracle - transformers| over the waveform."""
    from transformers import SpeechT5HifiGan, SpeechT5HifiGanConfig

    from oracle import hifigan_oracle as ho

    cfg = ho.HifiganConfig(initial_channel=initial_channel, upsample_rates=tuple(rates), upsample_kernels=tuple(kernels))
    sd = {k: v.double() for k, v in ho.make_state_dict(cfg, seed=seed).items()}
    model = SpeechT5HifiGan(SpeechT5HifiGanConfig(
        model_in_dim=cfg.n_mels, upsample_initial_channel=initial_channel, upsample_rates=list(rates),
        upsample_kernel_sizes=list(kernels), resblock_kernel_sizes=list(cfg.resblock_kernels),
        resblock_dilation_sizes=[list(d) for d in cfg.resblock_dilations], leaky_relu_slope=cfg.lrelu_slope,
        normalize_before=False)).double().eval()
    theirs = {k.replace("ups.", "upsampler."): v for k, v in sd.items()}
    own = model.state_dict()
    missing = [k for k in own if k not in theirs and k not in ("mean", "scale")]
    extra = [k for k in theirs if k not in own]
    assert not missing and not extra, (missing, extra)
    model.load_state_dict({**theirs, "mean": own["mean"], "scale": own["scale"]})
    mel = torch.randn(2, cfg.n_mels, frames, generator=torch.Generator().manual_seed(seed + 1), dtype=torch.float32).double()
    with torch.no_grad():
        want = model(mel.transpose(1, 2))
    got = ho.generator_forward(mel, sd, cfg)
    assert want.shape == got.shape, (want.shape, got.shape)
    return float((want - got).abs().max())


def witness_sanm_attention(b=2, t=70, lens=(70, 41), seed=21, layer="encoders.0"):
    """MultiHeadedAttentionSANM of the encoder oracle against two torch modules, in fp64, with the oracle's weights:
    ``nn.MultiheadAttention`` (packed in-projection = linear_q_k_v, out_proj = linear_out, key_padding_mask) for the
    softmax core and ``nn.Conv1d(512, 512, 11, padding=5, groups=512, bias=False)`` for the FSMN memory (plus its
    identity path and the masks).  -> (max |difference| on valid rows, max |FSMN part alone|)."""
    cfg = so.SanmConfig(num_blocks=2)
    sd = so.make_state_dict(cfg, seed=seed, dtype=torch.float64)
    d = cfg.output_size
    x = so.synth_feats(b, t, d, seed=seed + 1, dtype=torch.float64)
    ln = torch.tensor(lens, dtype=torch.int32)
    mask = (torch.arange(t)[None, :] < ln[:, None]).double()
    got = so._attention_sanm(x, mask, sd, layer, cfg)

    mha = torch.nn.MultiheadAttention(d, cfg.attention_heads, bias=True, batch_first=True, dtype=torch.float64).eval()
    conv = torch.nn.Conv1d(d, d, cfg.kernel_size, padding=(cfg.kernel_size - 1) // 2, groups=d, bias=False, dtype=torch.float64)
    with torch.no_grad():
        mha.in_proj_weight.copy_(sd[f"{layer}.self_attn.linear_q_k_v.weight"])
        mha.in_proj_bias.copy_(sd[f"{layer}.self_attn.linear_q_k_v.bias"])
        mha.out_proj.weight.copy_(sd[f"{layer}.self_attn.linear_out.weight"])
        mha.out_proj.bias.copy_(sd[f"{layer}.self_attn.linear_out.bias"])
        conv.weight.copy_(sd[f"{layer}.self_attn.fsmn_block.weight"])
        core, _ = mha(x, x, x, key_padding_mask=mask == 0, need_weights=False)
        v = torch.nn.functional.linear(x, sd[f"{layer}.self_attn.linear_q_k_v.weight"][2 * d:],
                                       sd[f"{layer}.self_attn.linear_q_k_v.bias"][2 * d:]) * mask[:, :, None]
        fsmn = (conv(v.transpose(1, 2)).transpose(1, 2) + v) * mask[:, :, None]
    valid = mask[:, :, None] > 0
    diff = torch.where(valid, got - (core + fsmn), torch.zeros_like(got))
    # scaled_dot_product_attention as a third voice for the core alone
    h, dk = cfg.attention_heads, d // cfg.attention_heads
    qkv = torch.nn.functional.linear(x, sd[f"{layer}.self_attn.linear_q_k_v.weight"], sd[f"{layer}.self_attn.linear_q_k_v.bias"])
    q, k, vv = (z.reshape(b, t, h, dk).transpose(1, 2) for z in torch.split(qkv, d, dim=-1))
    with torch.no_grad():
        ctx = torch.nn.functional.scaled_dot_product_attention(q, k, vv, attn_mask=(mask > 0)[:, None, None, :])
        sdpa = torch.nn.functional.linear(ctx.transpose(1, 2).reshape(b, t, d), sd[f"{layer}.self_attn.linear_out.weight"],
                                          sd[f"{layer}.self_attn.linear_out.bias"])
    diff2 = torch.where(valid, got - (sdpa + fsmn), torch.zeros_like(got))
    return max(float(diff.abs().max()), float(diff2.abs().max())), float(fsmn.abs().max())


def second_witnesses():
    e = witness_hifigan()
    print(f"hifigan oracle vs transformers.SpeechT5HifiGan (128 ch, 4x2): {e:.3e}")
    assert e < 1e-12, e
    e = witness_hifigan(initial_channel=64, rates=(8, 8, 2, 2), kernels=(16, 16, 4, 4), frames=12, seed=19)
    print(f"hifigan oracle vs transformers.SpeechT5HifiGan (64 ch, 8x8x2x2 = the deployed stage layout): {e:.3e}")
    assert e < 1e-12, e
    e, size = witness_sanm_attention()
    print(f"SAN-M attention oracle vs nn.MultiheadAttention / SDPA + nn.Conv1d(groups=512): {e:.3e} (FSMN part up to {size:.3f})")
    assert e < 1e-12 and size > 1e-3, (e, size)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "witness":
        second_witnesses()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "vocoder":
        witness_hifigan()
        vocoder_goldens()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "full":
        full_size_goldens()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "decoder":
        decoder_goldens()
        return
    worst = 0.0
    for kind, seed, n in FBANK_CASES:
        pcm = fo.synth_pcm(kind, n, seed)
        logmel = fo.kaldi_fbank(pcm)
        err = cross_check_fbank(pcm, logmel)
        worst = max(worst, err)
        assert err < 2e-4, f"{kind}/{seed}: oracle vs transformers kaldi-fbank differ by {err}"
        shift, scale = synth_cmvn(seed)
        feats = fo.apply_cmvn(fo.apply_lfr(logmel), shift, scale)
        np.savez_compressed(
            os.path.join(OUT, f"fbank_{kind}_s{seed}_{n}.npz"),
            pcm=pcm,
            logmel=logmel.astype(np.float32),
            feats=feats.astype(np.float32),
            shift=shift.astype(np.float32),
            scale=scale.astype(np.float32),
        )
        print(f"fbank {kind:8s} seed {seed} n {n:6d}: T={logmel.shape[0]:4d} T'={feats.shape[0]:3d} "
              f"max|oracle - transformers| = {err:.3e}")
    print(f"worst cross-implementation difference: {worst:.3e}")

    # encoder, tiny: full tensors
    cfg = so.SanmConfig(num_blocks=2)
    sd = so.make_state_dict(cfg, seed=0, dtype=torch.float64)
    feats = so.synth_feats(2, 40, 560, seed=11, dtype=torch.float64)
    lens = torch.tensor([40, 27], dtype=torch.int32)
    out, olens = so.encoder_forward(feats, lens, sd, cfg)
    np.savez_compressed(
        os.path.join(OUT, "encoder_tiny.npz"),
        feats=feats.numpy().astype(np.float32),
        lens=lens.numpy(),
        out=out.numpy().astype(np.float32),
        olens=olens.numpy(),
        seed=np.int32(0),
        num_blocks=np.int32(2),
    )
    print("encoder tiny:", tuple(out.shape), "abs max", float(out.abs().max()))

    # encoder, full depth: slices + checksums
    cfg = so.SanmConfig(num_blocks=50)
    sd = so.make_state_dict(cfg, seed=0, dtype=torch.float64)
    feats = so.synth_feats(2, 96, 560, seed=12, dtype=torch.float64)
    lens = torch.tensor([96, 61], dtype=torch.int32)
    out, olens = so.encoder_forward(feats, lens, sd, cfg)
    o = out.numpy()
    np.savez_compressed(
        os.path.join(OUT, "encoder_probe_50l.npz"),
        lens=lens.numpy(),
        rows_b0=o[0, [0, 1, 47, 95]].astype(np.float32),
        rows_b1=o[1, [0, 30, 60]].astype(np.float32),
        col_sums=o.sum(axis=(0, 1)).astype(np.float64),
        abs_mean=np.float64(np.abs(o).mean()),
        seed=np.int32(0),
        feats_seed=np.int32(12),
        num_blocks=np.int32(50),
    )
    print("encoder 50l:", tuple(out.shape), "abs mean", float(np.abs(o).mean()))


if __name__ == "__main__":
    main()
