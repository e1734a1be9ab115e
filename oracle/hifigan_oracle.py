"""CPU oracle for the HiFi-GAN generator (mel spectrogram -> waveform), torch CPU in fp32 or fp64.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package may import this module; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do, and only as the checker.

PARITY UNPINNED: the vocoder is not in the reference tree.  The reference reaches it through
``pipeline(task=Tasks.text_to_speech, model="damo/speech_sambert-hifigan_tts_zh-cn_16k")``
(/root/reference/tts-service/app/services/tts_service.py:34-38, 164-177) and the call
``pipeline_instance(input=seg, voice=voice, forward_params=...)`` (:617-625), i.e. inside ``modelscope==1.33.0`` /
``kantts==1.0.1`` (/root/reference/tts-service/requirements.txt:12-14), which are not installed here, ship no weights
and no test vectors, and cannot be fetched.  This file restates the published HiFi-GAN generator (Kong et al. 2020,
``Generator`` / ``ResBlock1`` of the public implementation): conv_pre(k 7) -> per upsampling stage
[leaky_relu 0.1 -> ConvTranspose1d(k, stride u, padding (k - u) / 2) -> mean over the stage's residual blocks (kernel
sizes 3 / 7 / 11, each three pairs of dilated (1, 3, 5) and plain convolutions with leaky_relu 0.1 in front of every
convolution and a residual add per pair)] -> leaky_relu (slope 0.01) -> conv_post(k 7) -> tanh.  Parameter names follow
that implementation with weight norm removed (``conv_pre``, ``ups.i``, ``resblocks.j.convs1.k``, ``resblocks.j.convs2.k``,
``conv_post``).  Whether the KAN-TTS build of the vocoder differs in details (causal padding, stage layout) could not be
checked here; the hyper-parameters are therefore all configurable.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import torch
import torch.nn.functional as F


@dataclass
class HifiganConfig:
    n_mels: int = 80
    initial_channel: int = 512
    upsample_rates: tuple = (8, 8, 2, 2)
    upsample_kernels: tuple = (16, 16, 4, 4)
    resblock_kernels: tuple = (3, 7, 11)
    resblock_dilations: tuple = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    lrelu_slope: float = 0.1
    final_slope: float = 0.01

    @property
    def hop(self) -> int:
        h = 1
        for u in self.upsample_rates:
            h *= u
        return h


def make_state_dict(cfg: HifiganConfig, seed: int = 0, dtype=torch.float32, gain: float = 1.0):
    """Seeded weights, scaled so that activations keep unit scale through the stack (He-style for leaky_relu)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)

    def rnd(*shape, fan_in):
        return (torch.randn(*shape, generator=g, dtype=torch.float32) * gain * (1.0 / fan_in) ** 0.5).to(dtype)

    sd = {}
    c0 = cfg.initial_channel
    sd["conv_pre.weight"] = rnd(c0, cfg.n_mels, 7, fan_in=cfg.n_mels * 7)
    sd["conv_pre.bias"] = rnd(c0, fan_in=16.0)
    ch = c0
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernels)):
        co = ch // 2
        sd[f"ups.{i}.weight"] = rnd(ch, co, k, fan_in=ch * k / u)   # ConvTranspose1d: [in, out, k]
        sd[f"ups.{i}.bias"] = rnd(co, fan_in=16.0)
        for j, (rk, dils) in enumerate(zip(cfg.resblock_kernels, cfg.resblock_dilations)):
            p = f"resblocks.{i * len(cfg.resblock_kernels) + j}"
            for m in range(len(dils)):
                sd[f"{p}.convs1.{m}.weight"] = rnd(co, co, rk, fan_in=co * rk * 4)
                sd[f"{p}.convs1.{m}.bias"] = rnd(co, fan_in=64.0)
                sd[f"{p}.convs2.{m}.weight"] = rnd(co, co, rk, fan_in=co * rk * 4)
                sd[f"{p}.convs2.{m}.bias"] = rnd(co, fan_in=64.0)
        ch = co
    sd["conv_post.weight"] = rnd(1, ch, 7, fan_in=ch * 7)
    sd["conv_post.bias"] = rnd(1, fan_in=16.0)
    return sd


def resblock(x, sd, p, kernel, dilations, slope):
    for m, d in enumerate(dilations):
        xt = F.leaky_relu(x, slope)
        xt = F.conv1d(xt, sd[f"{p}.convs1.{m}.weight"], sd[f"{p}.convs1.{m}.bias"], dilation=d, padding=(kernel * d - d) // 2)
        xt = F.leaky_relu(xt, slope)
        xt = F.conv1d(xt, sd[f"{p}.convs2.{m}.weight"], sd[f"{p}.convs2.{m}.bias"], padding=(kernel - 1) // 2)
        x = xt + x
    return x


@torch.no_grad()
def generator_forward(mel: torch.Tensor, sd, cfg: HifiganConfig) -> torch.Tensor:
    """mel (B, n_mels, T) -> waveform (B, T * hop) in (-1, 1)."""
    x = F.conv1d(mel, sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3)
    nk = len(cfg.resblock_kernels)
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernels)):
        x = F.leaky_relu(x, cfg.lrelu_slope)
        x = F.conv_transpose1d(x, sd[f"ups.{i}.weight"], sd[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        xs = None
        for j in range(nk):
            y = resblock(x, sd, f"resblocks.{i * nk + j}", cfg.resblock_kernels[j], cfg.resblock_dilations[j], cfg.lrelu_slope)
            xs = y if xs is None else xs + y
        x = xs / nk
    x = F.leaky_relu(x, cfg.final_slope)
    x = F.conv1d(x, sd["conv_post.weight"], sd["conv_post.bias"], padding=3)
    return torch.tanh(x).squeeze(1)
