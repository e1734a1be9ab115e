"""CPU oracle for the step behind the encoder: CIF predictor + Paraformer SAN-M decoder + greedy token choice.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package may import this module; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do, and only as the checker.

PARITY UNPINNED: this arithmetic is not in the reference tree.  The reference reaches it inside the ModelScope ASR
pipeline it builds at /root/reference/voice-service/app/services/voice_interface.py:832-841 and calls at :2049-2053
(model ``paraformer-large ... vocab8404``, :629-634, :776); what it consumes is the recognised text per sentence
(:2056-2067, :2108-2181).  ``funasr`` is unpinned and not vendored (/root/reference/voice-service/Dockerfile:25-36),
no weights, tests or golden vectors ship with the reference, and nothing of it can be run here.  This file restates
the published FunASR modules with the paraformer-large hyper-parameters:

* ``CifPredictorV2`` (idim 512, l_order 1, r_order 1, threshold 1.0, tail_threshold 0.45, smooth_factor 1.0,
  noise_threshold 0): ``alphas = relu(sigmoid(Linear(relu(Conv1d_k3(pad(h))))) * smooth - noise) * mask``, the tail
  process (0.45 added at the first position past the utterance, a zero frame appended, token count = floor(sum alphas))
  and ``cif`` (continuous integrate-and-fire over time with threshold 1.0);
* ``ParaformerSANMDecoder`` (16 blocks of ``DecoderLayerSANM`` = feed-forward with a LayerNorm on its hidden layer and a
  bias-free second linear, FSMN memory block (kernel 11) on the normalised feed-forward output plus residual, 4-head
  cross-attention onto the encoder output plus residual; then one feed-forward-only block WITHOUT residual,
  ``after_norm`` and the output layer onto the vocabulary), LayerNorm eps 1e-12;
* the greedy choice of ``Paraformer.inference``: arg-max over the vocabulary for the first ``token_num`` positions,
  ids of sos (1), eos (2) and blank (0) dropped.

Parameter names follow FunASR's ``state_dict`` (``predictor.*``, ``decoder.*``) so that a real ``model.pt`` can be
dropped in where one exists.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class DecoderConfig:
    encoder_size: int = 512
    attention_dim: int = 512
    attention_heads: int = 4
    linear_units: int = 2048
    num_blocks: int = 16          # att_layer_num == num_blocks: every block has FSMN + cross-attention
    kernel_size: int = 11
    sanm_shfit: int = 0
    vocab_size: int = 8404
    ln_eps: float = 1e-12
    # predictor
    l_order: int = 1
    r_order: int = 1
    threshold: float = 1.0
    tail_threshold: float = 0.45
    smooth_factor: float = 1.0
    noise_threshold: float = 0.0
    sos: int = 1
    eos: int = 2
    blank: int = 0


def make_state_dict(cfg: DecoderConfig, seed: int = 0, std: float = 0.02, dtype=torch.float32, alpha_bias: float = -0.4):
    """Seeded synthetic weights under FunASR's names.  ``alpha_bias`` shifts the predictor's output so that alphas sit
    around 0.4 (about one token per 2.5 frames), the density of real speech being lower still."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)

    def rnd(*shape, s=std):
        return (torch.randn(*shape, generator=g, dtype=torch.float32) * s).to(dtype)

    d, ff, e = cfg.attention_dim, cfg.linear_units, cfg.encoder_size
    sd = {}
    k = cfg.l_order + cfg.r_order + 1
    sd["predictor.cif_conv1d.weight"] = rnd(e, e, k)
    sd["predictor.cif_conv1d.bias"] = rnd(e)
    sd["predictor.cif_output.weight"] = rnd(1, e, s=0.05)
    sd["predictor.cif_output.bias"] = torch.full((1,), alpha_bias, dtype=dtype)

    def ffn(p):
        sd[f"{p}.norm1.weight"] = 1.0 + rnd(d, s=0.1)
        sd[f"{p}.norm1.bias"] = rnd(d, s=0.1)
        sd[f"{p}.feed_forward.w_1.weight"] = rnd(ff, d)
        sd[f"{p}.feed_forward.w_1.bias"] = rnd(ff)
        sd[f"{p}.feed_forward.norm.weight"] = 1.0 + rnd(ff, s=0.1)
        sd[f"{p}.feed_forward.norm.bias"] = rnd(ff, s=0.1)
        sd[f"{p}.feed_forward.w_2.weight"] = rnd(d, ff)

    for i in range(cfg.num_blocks):
        p = f"decoder.decoders.{i}"
        ffn(p)
        sd[f"{p}.norm2.weight"] = 1.0 + rnd(d, s=0.1)
        sd[f"{p}.norm2.bias"] = rnd(d, s=0.1)
        sd[f"{p}.self_attn.fsmn_block.weight"] = rnd(d, 1, cfg.kernel_size, s=0.1)
        sd[f"{p}.norm3.weight"] = 1.0 + rnd(d, s=0.1)
        sd[f"{p}.norm3.bias"] = rnd(d, s=0.1)
        sd[f"{p}.src_attn.linear_q.weight"] = rnd(d, d)
        sd[f"{p}.src_attn.linear_q.bias"] = rnd(d)
        sd[f"{p}.src_attn.linear_k_v.weight"] = rnd(2 * d, e)
        sd[f"{p}.src_attn.linear_k_v.bias"] = rnd(2 * d)
        sd[f"{p}.src_attn.linear_out.weight"] = rnd(d, d)
        sd[f"{p}.src_attn.linear_out.bias"] = rnd(d)
    ffn("decoder.decoders3.0")
    sd["decoder.after_norm.weight"] = 1.0 + rnd(d, s=0.1)
    sd["decoder.after_norm.bias"] = rnd(d, s=0.1)
    sd["decoder.output_layer.weight"] = rnd(cfg.vocab_size, d, s=0.05)
    sd["decoder.output_layer.bias"] = rnd(cfg.vocab_size, s=0.05)
    return sd


# ---- predictor -------------------------------------------------------------------------------------------
def predictor_alphas(hidden: torch.Tensor, lens: torch.Tensor, sd, cfg: DecoderConfig) -> torch.Tensor:
    """CifPredictorV2.forward up to the masked alphas: hidden (B, T, 512), lens (B,) -> alphas (B, T)."""
    b, t, _ = hidden.shape
    mask = (torch.arange(t)[None, :] < lens[:, None]).to(hidden.dtype)
    # rows past an utterance's length count as zeros (what a batch of one sees behind its last frame)
    h = hidden * mask[:, :, None]
    q = F.pad(h.transpose(1, 2), (cfg.l_order, cfg.r_order), value=0.0)
    out = torch.relu(F.conv1d(q, sd["predictor.cif_conv1d.weight"], sd["predictor.cif_conv1d.bias"])).transpose(1, 2)
    out = F.linear(out, sd["predictor.cif_output.weight"], sd["predictor.cif_output.bias"])
    alphas = torch.sigmoid(out)
    alphas = torch.relu(alphas * cfg.smooth_factor - cfg.noise_threshold)
    return alphas.squeeze(-1) * mask


def tail_process(hidden, alphas, lens, cfg: DecoderConfig):
    """CifPredictorV2.tail_process_fn with a mask: tail_threshold at position len[b], one zero frame appended."""
    b, t, d = hidden.shape
    mask = (torch.arange(t)[None, :] < lens[:, None]).to(alphas.dtype)
    zeros_t = torch.zeros((b, 1), dtype=alphas.dtype)
    ones_t = torch.ones_like(zeros_t)
    tail = (torch.cat([ones_t, mask], dim=1) - torch.cat([mask, zeros_t], dim=1)) * cfg.tail_threshold
    alphas = torch.cat([alphas, zeros_t], dim=1) + tail
    hidden = torch.cat([hidden, torch.zeros((b, 1, d), dtype=hidden.dtype)], dim=1)
    token_num = torch.floor(alphas.sum(dim=-1))
    return hidden, alphas, token_num


def cif(hidden, alphas, threshold: float):
    """funasr ``cif``: returns (acoustic_embeds (B, max fires, D) zero padded, fires (B, T), fire counts (B,))."""
    b, t, d = hidden.shape
    integrate = torch.zeros(b, dtype=hidden.dtype)
    frame = torch.zeros(b, d, dtype=hidden.dtype)
    fires, frames = [], []
    one = torch.ones(b, dtype=hidden.dtype)
    for i in range(t):
        alpha = alphas[:, i]
        completion = one - integrate
        integrate = integrate + alpha
        fires.append(integrate)
        fire = integrate >= threshold
        integrate = torch.where(fire, integrate - one, integrate)
        cur = torch.where(fire, completion, alpha)
        remainds = alpha - cur
        frame = frame + cur[:, None] * hidden[:, i, :]
        frames.append(frame)
        frame = torch.where(fire[:, None], remainds[:, None] * hidden[:, i, :], frame)
    fires = torch.stack(fires, 1)
    frames = torch.stack(frames, 1)
    outs, counts = [], []
    for k in range(b):
        idx = torch.nonzero(fires[k] >= threshold).reshape(-1)
        outs.append(frames[k, idx])
        counts.append(idx.numel())
    n = max(counts) if counts else 0
    emb = torch.zeros(b, n, d, dtype=hidden.dtype)
    for k in range(b):
        emb[k, : counts[k]] = outs[k]
    return emb, fires, torch.tensor(counts, dtype=torch.int32)


def cif_fire_frames(alphas, threshold: float, n_max: int, dtype=torch.float32):
    """The frame at which each token fires, from the integrate-and-fire recurrence of ``cif`` alone (float32 by default:
    the arithmetic the device runs): alphas (B, T + 1) with the tail weight in place -> (B, n_max) int32, frame index of
    token n's fire, -1 behind an utterance's fires.  With LFR frames of lfr_n * frame_shift = 60 ms a token spans
    (fire[n - 1], fire[n]] * 60 ms: the token time stamps the reference's pipeline reports per sentence
    (/root/reference/voice-service/app/services/voice_interface.py:2064-2067, 2124-2126)."""
    a = alphas.to(dtype)
    b, t = a.shape
    out = torch.full((b, n_max), -1, dtype=torch.int32)
    one = torch.ones((), dtype=dtype)
    thr = torch.tensor(threshold, dtype=dtype)
    for k in range(b):
        integrate = torch.zeros((), dtype=dtype)
        n = 0
        for i in range(t):
            integrate = integrate + a[k, i]
            if integrate >= thr:
                integrate = integrate - one
                if n < n_max:
                    out[k, n] = i
                n += 1
    return out


def predictor_forward(hidden, lens, sd, cfg: DecoderConfig):
    """-> (acoustic_embeds (B, N, 512), token_num (B,) int32, alphas (B, T + 1), fire counts (B,)).
    N = the largest token_num of the batch, as Paraformer crops it; token_num = floor(sum of alphas incl. the tail)."""
    mask = (torch.arange(hidden.shape[1])[None, :] < lens[:, None]).to(hidden.dtype)
    alphas = predictor_alphas(hidden, lens, sd, cfg)
    h, a, token_num = tail_process(hidden * mask[:, :, None], alphas, lens, cfg)
    emb, _, counts = cif(h, a, cfg.threshold)
    n = int(token_num.max().item()) if token_num.numel() else 0
    if emb.shape[1] < n:
        emb = F.pad(emb, (0, 0, 0, n - emb.shape[1]))
    return emb[:, :n], token_num.to(torch.int32), a, counts


# ---- decoder ---------------------------------------------------------------------------------------------
def _ffn(x, sd, p, cfg):
    """PositionwiseFeedForwardDecoderSANM: w_2(LayerNorm(relu(w_1 x))), w_2 without bias."""
    h = torch.relu(F.linear(x, sd[f"{p}.feed_forward.w_1.weight"], sd[f"{p}.feed_forward.w_1.bias"]))
    h = F.layer_norm(h, (cfg.linear_units,), sd[f"{p}.feed_forward.norm.weight"], sd[f"{p}.feed_forward.norm.bias"], cfg.ln_eps)
    return F.linear(h, sd[f"{p}.feed_forward.w_2.weight"])


def _fsmn(x, mask3, sd, p, cfg):
    """MultiHeadedAttentionSANMDecoder.forward (no cache): mask, depth-wise conv over time, + input, mask."""
    x = x * mask3
    left = (cfg.kernel_size - 1) // 2 + max(cfg.sanm_shfit, 0)
    right = cfg.kernel_size - 1 - left
    y = F.conv1d(F.pad(x.transpose(1, 2), (left, right)), sd[f"{p}.self_attn.fsmn_block.weight"], groups=x.shape[2])
    return (y.transpose(1, 2) + x) * mask3


def _cross_attention(x, memory, mem_mask, sd, p, cfg):
    """MultiHeadedAttentionCrossAtt.forward: 4-head attention of the tokens onto the encoder frames."""
    b, n, d = x.shape
    hh, dk = cfg.attention_heads, cfg.attention_dim // cfg.attention_heads
    q = F.linear(x, sd[f"{p}.src_attn.linear_q.weight"], sd[f"{p}.src_attn.linear_q.bias"])
    kv = F.linear(memory, sd[f"{p}.src_attn.linear_k_v.weight"], sd[f"{p}.src_attn.linear_k_v.bias"])
    k, v = torch.split(kv, hh * dk, dim=-1)
    qh = q.reshape(b, n, hh, dk).transpose(1, 2) * dk ** (-0.5)
    kh = k.reshape(b, -1, hh, dk).transpose(1, 2)
    vh = v.reshape(b, -1, hh, dk).transpose(1, 2)
    scores = torch.matmul(qh, kh.transpose(-2, -1))
    dead = (mem_mask == 0)[:, None, None, :]
    attn = torch.softmax(scores.masked_fill(dead, float("-inf")), dim=-1).masked_fill(dead, 0.0)
    ctx = torch.matmul(attn, vh).transpose(1, 2).reshape(b, n, hh * dk)
    return F.linear(ctx, sd[f"{p}.src_attn.linear_out.weight"], sd[f"{p}.src_attn.linear_out.bias"])


def decoder_layer(x, tok_mask3, memory, mem_mask, sd, p, cfg: DecoderConfig):
    """DecoderLayerSANM.forward, normalize_before, with self_attn (FSMN) and src_attn."""
    d = cfg.attention_dim
    residual = x
    t = _ffn(F.layer_norm(x, (d,), sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], cfg.ln_eps), sd, p, cfg)
    t = F.layer_norm(t, (d,), sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], cfg.ln_eps)
    x = residual + _fsmn(t, tok_mask3, sd, p, cfg)
    residual = x
    y = F.layer_norm(x, (d,), sd[f"{p}.norm3.weight"], sd[f"{p}.norm3.bias"], cfg.ln_eps)
    return residual + _cross_attention(y, memory, mem_mask, sd, p, cfg)


@torch.no_grad()
def decoder_forward(memory, mem_lens, embeds, token_num, sd, cfg: DecoderConfig):
    """ParaformerSANMDecoder.forward(hs_pad, hlens, ys_in_pad, ys_in_lens) -> logits (B, N, vocab); rows at or past
    token_num[b] are not meaningful (the reference computes them from padding) and are zeroed here."""
    b, n, d = embeds.shape
    tok_mask = (torch.arange(n)[None, :] < token_num[:, None]).to(embeds.dtype)
    mem_mask = (torch.arange(memory.shape[1])[None, :] < mem_lens[:, None]).to(embeds.dtype)
    x = embeds
    for i in range(cfg.num_blocks):
        x = decoder_layer(x, tok_mask[:, :, None], memory, mem_mask, sd, f"decoder.decoders.{i}", cfg)
    p = "decoder.decoders3.0"
    x = _ffn(F.layer_norm(x, (d,), sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], cfg.ln_eps), sd, p, cfg)  # no residual
    x = F.layer_norm(x, (d,), sd["decoder.after_norm.weight"], sd["decoder.after_norm.bias"], cfg.ln_eps)
    logits = F.linear(x, sd["decoder.output_layer.weight"], sd["decoder.output_layer.bias"])
    return logits * tok_mask[:, :, None]


@torch.no_grad()
def recognize(memory, mem_lens, sd, cfg: DecoderConfig):
    """Encoder output -> (token ids per utterance with sos / eos / blank dropped, logits, token_num, alphas)."""
    emb, token_num, alphas, _ = predictor_forward(memory, mem_lens, sd, cfg)
    mask = (torch.arange(memory.shape[1])[None, :] < mem_lens[:, None]).to(memory.dtype)
    logits = decoder_forward(memory * mask[:, :, None], mem_lens, emb, token_num, sd, cfg)
    ids = []
    for k in range(memory.shape[0]):
        y = logits[k, : int(token_num[k])].argmax(dim=-1).tolist()
        ids.append([t for t in y if t not in (cfg.sos, cfg.eos, cfg.blank)])
    return ids, logits, token_num, alphas
