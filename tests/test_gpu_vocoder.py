"""GPU parity of the HiFi-GAN generator (every convolution an implicit GEMM on the matrix cores), called through the C
ABI, against the CPU oracle (oracle/hifigan_oracle.py, fp64): waveform within 1e-2 of the oracle's (samples lie in
(-1, 1); 16-bit operands through up to 4 stages x 18 convolutions).  The reference holds no fixture for the vocoder
(parity unpinned)."""
import io
import wave

import numpy as np
import pytest
import torch

from oracle import hifigan_oracle as ho

pytestmark = pytest.mark.gpu

WAV_TOL = 1e-2


@pytest.fixture(scope="module")
def pkg():
    import toolbox_for_asr_and_tts_amd as p

    p.load()
    return p


def build(cfg, seed):
    from toolbox_for_asr_and_tts_amd.vocoder import HiFiGANGenerator
    sd = ho.make_state_dict(cfg, seed=seed)
    gen = HiFiGANGenerator(n_mels=cfg.n_mels, initial_channel=cfg.initial_channel, upsample_rates=cfg.upsample_rates,
                           upsample_kernels=cfg.upsample_kernels, resblock_kernels=cfg.resblock_kernels,
                           resblock_dilations=cfg.resblock_dilations, lrelu_slope=cfg.lrelu_slope, final_slope=cfg.final_slope)
    gen.load_state_dict(sd)
    return gen, sd


def check(gen, sd, cfg, mel):
    want = ho.generator_forward(mel.double(), {k: v.double() for k, v in sd.items()}, cfg)
    got = gen.forward(mel.cuda())
    torch.cuda.synchronize()
    assert got.shape == want.shape == (mel.shape[0], mel.shape[2] * cfg.hop)
    assert bool(torch.isfinite(got).all())
    err = float((got.cpu().double() - want).abs().max())
    assert err <= WAV_TOL, err
    assert float(want.abs().max()) > 0.05  # a signal, not silence
    return err


@pytest.mark.parametrize("cfg,t_mel", [
    (ho.HifiganConfig(initial_channel=128, upsample_rates=(4, 2), upsample_kernels=(8, 4)), 37),       # two stages, odd length
    (ho.HifiganConfig(initial_channel=256, upsample_rates=(5, 4, 2), upsample_kernels=(15, 8, 4),
                      resblock_kernels=(3, 5), resblock_dilations=((1, 2, 4), (1, 3, 5))), 19),          # k = 3 u (three taps per phase)
    (ho.HifiganConfig(), 24),                                                                            # the published V1 layout
])
def test_generator_matches_oracle(pkg, cfg, t_mel):
    gen, sd = build(cfg, seed=3)
    g = torch.Generator().manual_seed(t_mel)
    mel = torch.randn(2, cfg.n_mels, t_mel, generator=g)
    err = check(gen, sd, cfg, mel)
    print(f"channels {cfg.initial_channel}, rates {cfg.upsample_rates}: max waveform error {err:.3e}")


@pytest.mark.parametrize("cfg,t_mel", [
    # one second of the published layout: 752 / 6016 / 12032 / 24064 positions per stage, i.e. every tile shape the
    # vocoder configuration runs with, many position tiles per item (taps reaching across tile borders, dilation 5)
    (ho.HifiganConfig(), 94),
    # a short generator on a long spectrogram: the large position tiles of every channel width (128, 64, 32 channels),
    # the last tile of each stage partly past the end
    (ho.HifiganConfig(initial_channel=128, upsample_rates=(2, 2), upsample_kernels=(4, 4)), 701),
    # 256 -> 128 channels at 96 mel bins: input channels that are no multiple of 64 (one workgroup per CU shapes)
    (ho.HifiganConfig(n_mels=96, initial_channel=256, upsample_rates=(4, 4), upsample_kernels=(8, 8), resblock_kernels=(3, 11),
                      resblock_dilations=((1, 3, 5), (1, 3, 5))), 150),
])
def test_generator_long_inputs_match_oracle(pkg, cfg, t_mel):
    gen, sd = build(cfg, seed=11)
    mel = torch.randn(1, cfg.n_mels, t_mel, generator=torch.Generator().manual_seed(t_mel))
    err = check(gen, sd, cfg, mel)
    print(f"channels {cfg.initial_channel}, {t_mel} frames: max waveform error {err:.3e}")


@pytest.mark.parametrize("cfg,t_mel", [
    (ho.HifiganConfig(), 94),                                                                            # 64- and 32-channel stages fused
    (ho.HifiganConfig(initial_channel=128, upsample_rates=(2, 2), upsample_kernels=(4, 4)), 701),     # both stages fused
    (ho.HifiganConfig(initial_channel=64, upsample_rates=(3,), upsample_kernels=(9,), resblock_kernels=(5, 13),
                      resblock_dilations=((1, 2, 7), (2, 1, 4))), 333),                                # other widths and dilations
])
def test_fused_pairs_equal_the_two_launch_form(pkg, cfg, t_mel, monkeypatch):
    """Stages of 32 and 64 channels run a residual pair as one launch; its arithmetic (operand rounding, accumulation
    order) is that of the two launches, so the waveforms are equal bit for bit."""
    gen, sd = build(cfg, seed=17)
    mel = torch.randn(3, cfg.n_mels, t_mel, generator=torch.Generator().manual_seed(t_mel + 1)).cuda()
    monkeypatch.setenv("ASRFE_VOC_NO_FUSE", "1")
    two = gen.forward(mel).clone()
    monkeypatch.setenv("ASRFE_VOC_NO_FUSE", "0")
    one = gen.forward(mel).clone()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(one).all()) and float(one.abs().max()) > 0.05
    assert torch.equal(one, two)


@pytest.mark.parametrize("t_mel", [1, 2, 3])
def test_generator_shortest_inputs(pkg, t_mel):
    """One to three mel frames: every stage is shorter than a position tile and, in the fused stages, than the taps' reach."""
    cfg = ho.HifiganConfig()
    gen, sd = build(cfg, seed=23)
    mel = torch.randn(2, cfg.n_mels, t_mel, generator=torch.Generator().manual_seed(100 + t_mel))
    check(gen, sd, cfg, mel)


def test_generator_fixture(pkg):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "vocoder_tiny.npz"))
    cfg = ho.HifiganConfig(initial_channel=int(g["initial_channel"]), upsample_rates=tuple(g["rates"].tolist()),
                           upsample_kernels=tuple(g["kernels"].tolist()))
    gen, _ = build(cfg, seed=int(g["seed"]))
    got = gen.forward(torch.from_numpy(g["mel"]).cuda())
    assert np.max(np.abs(got.cpu().numpy() - g["wav"])) <= WAV_TOL


def test_generator_batch_128_properties(pkg):
    """The batch size of the vocoder configuration (128): every item equals the same item run alone (bit for bit),
    edge frames included; one item against the oracle."""
    cfg = ho.HifiganConfig()
    gen, sd = build(cfg, seed=5)
    g = torch.Generator().manual_seed(9)
    mel = torch.randn(128, 80, 12, generator=g)
    wav = gen.forward(mel.cuda())
    torch.cuda.synchronize()
    assert wav.shape == (128, 12 * 256) and bool(torch.isfinite(wav).all())
    for k in (0, 77, 127):
        alone = gen.forward(mel[k:k + 1].cuda())
        assert torch.equal(alone[0], wav[k])
    want = ho.generator_forward(mel[77:78].double(), {k: v.double() for k, v in sd.items()}, cfg)
    assert float((wav[77].cpu().double() - want[0]).abs().max()) <= WAV_TOL


def test_generator_at_the_vocoder_configurations_shape(pkg):
    """128 items of 94 mel frames (one second each), the shape the vocoder rate is quoted on, as one call: items 0, 63
    and 127 equal the same item run alone bit for bit (no item sees its neighbours through a position tile that spans
    two of them), items 63 and 127 are within 1e-2 of the oracle, and the per-item sums add up to the batch's sum."""
    cfg = ho.HifiganConfig()
    gen, sd = build(cfg, seed=29)
    mel = torch.randn(128, 80, 94, generator=torch.Generator().manual_seed(30))
    wav = gen.forward(mel.cuda()).clone()
    torch.cuda.synchronize()
    assert wav.shape == (128, 94 * 256) and bool(torch.isfinite(wav).all())
    for k in (0, 63, 127):
        alone = gen.forward(mel[k:k + 1].cuda())
        assert torch.equal(alone[0], wav[k]), k
    sd64 = {k: v.double() for k, v in sd.items()}
    want = ho.generator_forward(mel[[63, 127]].double(), sd64, cfg)
    err = float((wav[[63, 127]].cpu().double() - want).abs().max())
    assert err <= WAV_TOL and float(want.abs().max()) > 0.05, err
    per_item = wav.double().sum(dim=1)
    assert abs(float(per_item.sum()) - float(wav.double().sum())) <= 1e-9 * float(wav.double().abs().sum())
    print(f"128 x 94 frames: max waveform error of items 63, 127: {err:.3e}")


def test_weight_norm_pairs_and_wav_bytes(pkg):
    """A state_dict with weight norm still in place loads to the same generator; wav_bytes is a 16-bit mono WAV image."""
    from toolbox_for_asr_and_tts_amd.vocoder import HiFiGANGenerator
    cfg = ho.HifiganConfig(initial_channel=128, upsample_rates=(4, 2), upsample_kernels=(8, 4))
    sd = ho.make_state_dict(cfg, seed=6)
    wn = {}
    for k, v in sd.items():
        if k.endswith(".weight"):
            norm = v.flatten(1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
            wn["generator." + k[:-7] + ".weight_g"] = norm.clone()
            wn["generator." + k[:-7] + ".weight_v"] = v * 3.0      # any positive multiple of the direction
        else:
            wn["generator." + k] = v
    kw = dict(initial_channel=128, upsample_rates=(4, 2), upsample_kernels=(8, 4), sample_rate=16000)
    a = HiFiGANGenerator(**kw).load_state_dict(sd)
    b = HiFiGANGenerator(**kw).load_state_dict(wn)
    mel = torch.randn(1, 80, 10, generator=torch.Generator().manual_seed(1)).cuda()
    wa, wb = a.forward(mel), b.forward(mel)
    assert float((wa - wb).abs().max()) <= 2e-3      # g v / ||v|| re-rounds the weights: not bit-identical
    img = a.wav_bytes(wa[0])
    with wave.open(io.BytesIO(img), "rb") as wf:
        assert (wf.getnchannels(), wf.getsampwidth(), wf.getframerate(), wf.getnframes()) == (1, 2, 16000, 80)
        pcm = np.frombuffer(wf.readframes(80), dtype=np.int16)
    assert np.array_equal(pcm, (wa[0].cpu().clamp(-1, 1) * 32767).round().to(torch.int16).numpy())


def test_vocoder_refusals(pkg):
    from toolbox_for_asr_and_tts_amd.vocoder import HiFiGANGenerator
    cfg = ho.HifiganConfig(initial_channel=128, upsample_rates=(4, 2), upsample_kernels=(8, 4))
    sd = ho.make_state_dict(cfg, seed=7)
    with pytest.raises(pkg.AsrfeError, match="multiple of its stride"):
        HiFiGANGenerator(initial_channel=128, upsample_rates=(4, 2), upsample_kernels=(7, 4)).load_state_dict(sd)
    with pytest.raises(pkg.AsrfeError, match="multiple of 32"):
        HiFiGANGenerator(initial_channel=96, upsample_rates=(4, 2), upsample_kernels=(8, 4)).load_state_dict(sd)
    bad = {k: v for k, v in sd.items() if k != "ups.1.bias"}
    with pytest.raises(pkg.AsrfeError, match="missing tensor ups.1.bias"):
        HiFiGANGenerator(initial_channel=128, upsample_rates=(4, 2), upsample_kernels=(8, 4)).load_state_dict(bad)
    gen = HiFiGANGenerator(initial_channel=128, upsample_rates=(4, 2), upsample_kernels=(8, 4))
    with pytest.raises(ValueError):
        gen.workspace_bytes(1, 10)                               # nothing loaded
    gen.load_state_dict(sd)
    with pytest.raises(ValueError):
        gen.forward(torch.zeros(1, 80, 10))                      # host tensor
