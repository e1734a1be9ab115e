"""Seeded synthetic inputs for benchmarks: 16 kHz int16 PCM clips and random-init encoder weights.

No dataset or checkpoint is reachable offline, so throughput is measured on synthetic data of the real
shape: Gaussian PCM (sigma 3000, speech-like crest) whose first 10 % is exact zeros (log-floor frames),
and N(0, 0.02) weights / LayerNorm gamma 1, beta 0 under FunASR's ``SANMEncoder`` state_dict names.
"""
from __future__ import annotations

import numpy as np
import torch


def bench_pcm(batch: int, n_samples: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    x = rng.normal(0.0, 3000.0, size=(batch, n_samples)).astype(np.float32)
    x[:, : n_samples // 10] = 0.0
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def encoder_state_dict(num_blocks: int = 50, input_size: int = 560, d: int = 512, ff: int = 2048, kernel: int = 11,
                       seed: int = 0, std: float = 0.02):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)

    def rnd(*shape):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * std

    sd = {}
    for li in range(num_blocks):
        p = "encoders0.0" if li == 0 else f"encoders.{li - 1}"
        din = input_size if li == 0 else d
        sd[f"{p}.norm1.weight"] = torch.ones(din)
        sd[f"{p}.norm1.bias"] = torch.zeros(din)
        sd[f"{p}.self_attn.linear_q_k_v.weight"] = rnd(3 * d, din)
        sd[f"{p}.self_attn.linear_q_k_v.bias"] = rnd(3 * d)
        sd[f"{p}.self_attn.fsmn_block.weight"] = rnd(d, 1, kernel)
        sd[f"{p}.self_attn.linear_out.weight"] = rnd(d, d)
        sd[f"{p}.self_attn.linear_out.bias"] = rnd(d)
        sd[f"{p}.norm2.weight"] = torch.ones(d)
        sd[f"{p}.norm2.bias"] = torch.zeros(d)
        sd[f"{p}.feed_forward.w_1.weight"] = rnd(ff, d)
        sd[f"{p}.feed_forward.w_1.bias"] = rnd(ff)
        sd[f"{p}.feed_forward.w_2.weight"] = rnd(d, ff)
        sd[f"{p}.feed_forward.w_2.bias"] = rnd(d)
    sd["after_norm.weight"] = torch.ones(d)
    sd["after_norm.bias"] = torch.zeros(d)
    return sd


def decoder_state_dict(num_blocks: int = 16, d: int = 512, ff: int = 2048, kernel: int = 11, vocab: int = 8404,
                       seed: int = 1, std: float = 0.02, alpha_bias: float = -0.4):
    """Random-init CIF predictor + Paraformer decoder under FunASR's ``state_dict`` names (``predictor.*``,
    ``decoder.*``).  ``alpha_bias`` centres the firing weights near 0.4, i.e. about 200 tokens per 30 s clip."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)

    def rnd(*shape, s=std):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * s

    sd = {
        "predictor.cif_conv1d.weight": rnd(d, d, 3),
        "predictor.cif_conv1d.bias": rnd(d),
        "predictor.cif_output.weight": rnd(1, d, s=0.05),
        "predictor.cif_output.bias": torch.full((1,), alpha_bias),
    }

    def ffn(p):
        sd[f"{p}.norm1.weight"] = torch.ones(d)
        sd[f"{p}.norm1.bias"] = torch.zeros(d)
        sd[f"{p}.feed_forward.w_1.weight"] = rnd(ff, d)
        sd[f"{p}.feed_forward.w_1.bias"] = rnd(ff)
        sd[f"{p}.feed_forward.norm.weight"] = torch.ones(ff)
        sd[f"{p}.feed_forward.norm.bias"] = torch.zeros(ff)
        sd[f"{p}.feed_forward.w_2.weight"] = rnd(d, ff)

    for i in range(num_blocks):
        p = f"decoder.decoders.{i}"
        ffn(p)
        for nm in ("norm2", "norm3"):
            sd[f"{p}.{nm}.weight"] = torch.ones(d)
            sd[f"{p}.{nm}.bias"] = torch.zeros(d)
        sd[f"{p}.self_attn.fsmn_block.weight"] = rnd(d, 1, kernel, s=0.1)
        sd[f"{p}.src_attn.linear_q.weight"] = rnd(d, d)
        sd[f"{p}.src_attn.linear_q.bias"] = rnd(d)
        sd[f"{p}.src_attn.linear_k_v.weight"] = rnd(2 * d, d)
        sd[f"{p}.src_attn.linear_k_v.bias"] = rnd(2 * d)
        sd[f"{p}.src_attn.linear_out.weight"] = rnd(d, d)
        sd[f"{p}.src_attn.linear_out.bias"] = rnd(d)
    ffn("decoder.decoders3.0")
    sd["decoder.after_norm.weight"] = torch.ones(d)
    sd["decoder.after_norm.bias"] = torch.zeros(d)
    sd["decoder.output_layer.weight"] = rnd(vocab, d, s=0.05)
    sd["decoder.output_layer.bias"] = rnd(vocab, s=0.05)
    return sd


V1_RATES, V1_KERNELS, V1_RES_KERNELS = (8, 8, 2, 2), (16, 16, 4, 4), (3, 7, 11)


def vocoder_state_dict(initial_channel: int = 512, n_mels: int = 80, rates=V1_RATES, kernels=V1_KERNELS,
                       res_kernels=V1_RES_KERNELS, seed: int = 2):
    """Random-init HiFi-GAN generator (published V1 layout by default) under its ``state_dict`` names, weight norm
    removed; scaled so that activations keep unit scale through the stack."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)

    def rnd(*shape, fan_in):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * (1.0 / fan_in) ** 0.5

    sd = {"conv_pre.weight": rnd(initial_channel, n_mels, 7, fan_in=n_mels * 7), "conv_pre.bias": rnd(initial_channel, fan_in=16.0)}
    ch = initial_channel
    for i, (u, k) in enumerate(zip(rates, kernels)):
        co = ch // 2
        sd[f"ups.{i}.weight"] = rnd(ch, co, k, fan_in=ch * k / u)
        sd[f"ups.{i}.bias"] = rnd(co, fan_in=16.0)
        for j, rk in enumerate(res_kernels):
            p = f"resblocks.{i * len(res_kernels) + j}"
            for m in range(3):
                for c in ("convs1", "convs2"):
                    sd[f"{p}.{c}.{m}.weight"] = rnd(co, co, rk, fan_in=co * rk * 4)
                    sd[f"{p}.{c}.{m}.bias"] = rnd(co, fan_in=64.0)
        ch = co
    sd["conv_post.weight"] = rnd(1, ch, 7, fan_in=ch * 7)
    sd["conv_post.bias"] = rnd(1, fan_in=16.0)
    return sd


def vocoder_flops(t_mel: int, initial_channel: int = 512, n_mels: int = 80, rates=V1_RATES, kernels=V1_KERNELS,
                  res_kernels=V1_RES_KERNELS) -> float:
    """Multiply-add FLOPs (2 per MAC) of one item's convolutions: conv_pre, the transposed convolutions, the residual
    blocks (six convolutions per block) and conv_post."""
    fl = 2.0 * t_mel * initial_channel * n_mels * 7
    ch, t = initial_channel, t_mel
    for u, k in zip(rates, kernels):
        co = ch // 2
        t *= u
        fl += 2.0 * t * co * ch * (k // u)
        for rk in res_kernels:
            fl += 6 * 2.0 * t * co * co * rk
        ch = co
    return fl + 2.0 * t * ch * 7


def vad_state_dict(input_dim: int = 400, input_affine_dim: int = 140, fsmn_layers: int = 4, linear_dim: int = 250,
                   proj_dim: int = 128, lorder: int = 20, output_affine_dim: int = 140, output_dim: int = 248,
                   seed: int = 0) -> dict:
    """Random-init weights of the FSMN voice-activity network under FunASR's names (``encoder.*``)."""
    g = torch.Generator().manual_seed(seed)

    def lin(n, k):
        return torch.randn(n, k, generator=g) / float(np.sqrt(k))

    sd = {"encoder.in_linear1.linear.weight": lin(input_affine_dim, input_dim),
          "encoder.in_linear1.linear.bias": 0.1 * torch.randn(input_affine_dim, generator=g),
          "encoder.in_linear2.linear.weight": lin(linear_dim, input_affine_dim),
          "encoder.in_linear2.linear.bias": 0.1 * torch.randn(linear_dim, generator=g)}
    for i in range(fsmn_layers):
        p = f"encoder.fsmn.{i}"
        sd[p + ".linear.linear.weight"] = lin(proj_dim, linear_dim)
        sd[p + ".fsmn_block.conv_left.weight"] = 0.1 * torch.randn(proj_dim, 1, lorder, 1, generator=g)
        sd[p + ".affine.linear.weight"] = lin(linear_dim, proj_dim)
        sd[p + ".affine.linear.bias"] = 0.1 * torch.randn(linear_dim, generator=g)
    sd["encoder.out_linear1.linear.weight"] = lin(output_affine_dim, linear_dim)
    sd["encoder.out_linear1.linear.bias"] = 0.1 * torch.randn(output_affine_dim, generator=g)
    sd["encoder.out_linear2.linear.weight"] = lin(output_dim, output_affine_dim)
    sd["encoder.out_linear2.linear.bias"] = 0.1 * torch.randn(output_dim, generator=g)
    return sd
