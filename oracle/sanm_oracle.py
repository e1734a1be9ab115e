"""CPU oracle for the Paraformer SAN-M encoder forward (torch CPU, fp32 or fp64).

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker.

PARITY UNPINNED: the encoder arithmetic is not in the reference tree.  The
reference reaches it through ``funasr.AutoModel`` / the ModelScope ASR pipeline
(/root/reference/voice-service/app/services/voice_interface.py:832-841 builds
it, :2049-2053 calls it; the model is paraformer-large, 16 kHz, vocab 8404,
voice_interface.py:629-634,776), ``funasr`` is unpinned and not vendored
(/root/reference/voice-service/Dockerfile:25-36) and no weights, tests or
golden vectors ship with the reference.  This file restates the published
FunASR modules ``SANMEncoder``, ``EncoderLayerSANM``,
``MultiHeadedAttentionSANM``, ``PositionwiseFeedForward``,
``SinusoidalPositionEncoder`` and its ``LayerNorm`` (eps 1e-12) with the
paraformer-large hyper-parameters (input 560, d_model 512, 4 heads, FFN 2048,
50 blocks, FSMN kernel 11, sanm_shfit 0, normalize_before).  Parameter names
follow FunASR's ``state_dict`` so a real ``model.pt`` (keys under ``encoder.``)
can be dropped in where one exists.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class SanmConfig:
    input_size: int = 560
    output_size: int = 512
    attention_heads: int = 4
    linear_units: int = 2048
    num_blocks: int = 50
    kernel_size: int = 11
    sanm_shfit: int = 0
    ln_eps: float = 1e-12


def layer_prefixes(cfg: SanmConfig):
    return ["encoders0.0"] + [f"encoders.{i}" for i in range(cfg.num_blocks - 1)]


def make_state_dict(cfg: SanmConfig, seed: int = 0, std: float = 0.02, dtype=torch.float32):
    """Seeded synthetic weights: N(0, std) for every weight and bias, LayerNorm gamma=1, beta=0.

    Uses a private CPU generator so the numbers depend only on (cfg, seed, torch version).
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)

    def rnd(*shape):
        return (torch.randn(*shape, generator=g, dtype=torch.float32) * std).to(dtype)

    sd = {}
    d, ff = cfg.output_size, cfg.linear_units
    for li, p in enumerate(layer_prefixes(cfg)):
        din = cfg.input_size if li == 0 else d
        sd[f"{p}.norm1.weight"] = torch.ones(din, dtype=dtype)
        sd[f"{p}.norm1.bias"] = torch.zeros(din, dtype=dtype)
        sd[f"{p}.self_attn.linear_q_k_v.weight"] = rnd(3 * d, din)
        sd[f"{p}.self_attn.linear_q_k_v.bias"] = rnd(3 * d)
        sd[f"{p}.self_attn.fsmn_block.weight"] = rnd(d, 1, cfg.kernel_size)
        sd[f"{p}.self_attn.linear_out.weight"] = rnd(d, d)
        sd[f"{p}.self_attn.linear_out.bias"] = rnd(d)
        sd[f"{p}.norm2.weight"] = torch.ones(d, dtype=dtype)
        sd[f"{p}.norm2.bias"] = torch.zeros(d, dtype=dtype)
        sd[f"{p}.feed_forward.w_1.weight"] = rnd(ff, d)
        sd[f"{p}.feed_forward.w_1.bias"] = rnd(ff)
        sd[f"{p}.feed_forward.w_2.weight"] = rnd(d, ff)
        sd[f"{p}.feed_forward.w_2.bias"] = rnd(d)
    sd["after_norm.weight"] = torch.ones(d, dtype=dtype)
    sd["after_norm.bias"] = torch.zeros(d, dtype=dtype)
    return sd


def sinusoidal_pe(t: int, depth: int, dtype=torch.float32) -> torch.Tensor:
    """FunASR SinusoidalPositionEncoder: positions 1..T, [sin | cos], increment ln(1e4)/(depth/2 - 1)."""
    pos = torch.arange(1, t + 1, dtype=dtype)
    inc = math.log(10000.0) / (depth / 2 - 1)
    inv = torch.exp(torch.arange(depth // 2, dtype=dtype) * (-inc))
    st = pos[:, None] * inv[None, :]
    return torch.cat([torch.sin(st), torch.cos(st)], dim=1)


def _attention_sanm(x, mask, sd, p, cfg: SanmConfig):
    """MultiHeadedAttentionSANM.forward: softmax attention plus the FSMN memory block on V."""
    b, t, _ = x.shape
    h = cfg.attention_heads
    d = cfg.output_size
    dk = d // h
    qkv = F.linear(x, sd[f"{p}.self_attn.linear_q_k_v.weight"], sd[f"{p}.self_attn.linear_q_k_v.bias"])
    q, k, v = torch.split(qkv, d, dim=-1)
    # forward_fsmn
    m3 = mask.reshape(b, t, 1).to(x.dtype)
    vin = v * m3
    left = (cfg.kernel_size - 1) // 2 + max(cfg.sanm_shfit, 0)
    right = cfg.kernel_size - 1 - left
    xc = F.pad(vin.transpose(1, 2), (left, right), value=0.0)
    xc = F.conv1d(xc, sd[f"{p}.self_attn.fsmn_block.weight"], bias=None, groups=d)
    fsmn = (xc.transpose(1, 2) + vin) * m3
    # attention
    qh = q.reshape(b, t, h, dk).transpose(1, 2) * dk ** (-0.5)
    kh = k.reshape(b, t, h, dk).transpose(1, 2)
    vh = v.reshape(b, t, h, dk).transpose(1, 2)
    scores = torch.matmul(qh, kh.transpose(-2, -1))
    kmask = (mask.reshape(b, 1, 1, t) == 0)
    scores = scores.masked_fill(kmask, float("-inf"))
    attn = torch.softmax(scores, dim=-1).masked_fill(kmask, 0.0)
    ctx = torch.matmul(attn, vh).transpose(1, 2).reshape(b, t, d)
    out = F.linear(ctx, sd[f"{p}.self_attn.linear_out.weight"], sd[f"{p}.self_attn.linear_out.bias"])
    return out + fsmn


def encoder_layer(x, mask, sd, p, cfg: SanmConfig, in_size: int):
    """EncoderLayerSANM.forward with normalize_before=True, concat_after=False, dropout off."""
    residual = x
    x = F.layer_norm(x, (in_size,), sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], cfg.ln_eps)
    att = _attention_sanm(x, mask, sd, p, cfg)
    x = residual + att if in_size == cfg.output_size else att
    residual = x
    x = F.layer_norm(x, (cfg.output_size,), sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], cfg.ln_eps)
    x = F.linear(x, sd[f"{p}.feed_forward.w_1.weight"], sd[f"{p}.feed_forward.w_1.bias"])
    x = F.linear(torch.relu(x), sd[f"{p}.feed_forward.w_2.weight"], sd[f"{p}.feed_forward.w_2.bias"])
    return residual + x


@torch.no_grad()
def encoder_forward(feats: torch.Tensor, lens: torch.Tensor, sd, cfg: SanmConfig):
    """SANMEncoder.forward(xs_pad (B, T', 560), ilens (B,)) -> (xs (B, T', 512), olens (B,)).

    Rows at or past a sequence's length are not meaningful in the reference (they are computed
    from padding); they are zeroed here so that comparisons only see valid rows.
    """
    b, t, _ = feats.shape
    dtype = feats.dtype
    mask = (torch.arange(t)[None, :] < lens[:, None]).to(dtype)  # (B, T) 1 = valid
    x = feats * cfg.output_size ** 0.5
    x = x + sinusoidal_pe(t, cfg.input_size, dtype)[None]
    for li, p in enumerate(layer_prefixes(cfg)):
        x = encoder_layer(x, mask, sd, p, cfg, cfg.input_size if li == 0 else cfg.output_size)
    x = F.layer_norm(x, (cfg.output_size,), sd["after_norm.weight"], sd["after_norm.bias"], cfg.ln_eps)
    x = torch.where(mask[:, :, None] > 0, x, torch.zeros_like(x))
    return x, mask.sum(1).to(torch.int32)


def synth_feats(b: int, t: int, dim: int, seed: int, dtype=torch.float32):
    """Seeded feature tensor standing in for CMVN-normalised LFR features (unit variance)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.randn(b, t, dim, generator=g, dtype=torch.float32).to(dtype)
