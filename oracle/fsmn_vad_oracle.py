"""CPU oracle for the FSMN voice-activity network and the streaming frontend in front of it.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package may import this module; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do, and only as the checker.

PARITY UNPINNED: this arithmetic is not in the reference tree.  The reference loads the model
``speech_fsmn_vad_zh-cn-16k-common-pytorch`` through ``funasr.AutoModel``
(/root/reference/voice-service/app/services/voice_interface.py:723-727) and calls it once per 400 ms chunk of a
streaming session, ``vad_model.generate(input=audio_np, cache=self.vad_cache, is_final=False, chunk_size=...)``
(:1585-1590), reading ``[{"value": [...]}]`` (:1593-1600).  ``funasr`` is unpinned and not vendored, no weights, tests
or golden vectors ship with the reference, and nothing of it can be run here.  This file restates the published FunASR
modules with that model's hyper-parameters (config.yaml of the model directory):

* ``WavFrontendOnline`` (fs 16000, Hamming, 80 mel, 25 / 10 ms, lfr_m 5, lfr_n 1, dither 0): framing over the
  concatenation of the samples left over from earlier chunks and the new chunk; LFR with ``(lfr_m - 1) // 2`` copies of
  the stream's first frame in front, rows emitted as soon as their right context exists, the rest at ``is_final`` with
  the last frame repeated;
* the ``FSMN`` encoder: ``in_linear1`` (400 -> 140), ``in_linear2`` (140 -> 250), ReLU, 4 blocks of
  [``linear`` 250 -> 128 without bias, ``fsmn_block``: out[t] = p[t] + sum_j w[c, j] p[t - 19 + j] over 20 causal taps
  (a depthwise Conv2d with kernel [20, 1]; the 19 rows before the chunk come from the cache), ``affine`` 128 -> 250,
  ReLU], ``out_linear1`` (250 -> 140), ``out_linear2`` (140 -> 248), softmax;
* the per-frame level the detector reads beside the scores: 10 log10(sum of squares of the frame's 400 samples in
  [-1, 1] + 1e-6).

Parameter names follow FunASR's ``state_dict`` (``encoder.in_linear1.linear.weight`` ...).  The detector's state
machine (``E2EVadModel``) is host logic of the package itself (``vad.py``) and is tested against hand-derived
segment boundaries.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import fbank_oracle


@dataclass
class VadNetConfig:
    input_dim: int = 400
    input_affine_dim: int = 140
    fsmn_layers: int = 4
    linear_dim: int = 250
    proj_dim: int = 128
    lorder: int = 20
    output_affine_dim: int = 140
    output_dim: int = 248


def make_state_dict(cfg: VadNetConfig = VadNetConfig(), seed: int = 0, sil_bias: float = 0.0) -> dict:
    """Seeded synthetic weights under FunASR's names (float32 numpy).  Scales are chosen so that the softmax is neither
    flat nor saturated; ``sil_bias`` shifts the silence class (pdf 0)."""
    rng = np.random.default_rng(seed)

    def lin(n, k, s):
        return (rng.standard_normal((n, k)) * s / np.sqrt(k)).astype(np.float32)

    sd = {}
    sd["encoder.in_linear1.linear.weight"] = lin(cfg.input_affine_dim, cfg.input_dim, 1.0)
    sd["encoder.in_linear1.linear.bias"] = (rng.standard_normal(cfg.input_affine_dim) * 0.1).astype(np.float32)
    sd["encoder.in_linear2.linear.weight"] = lin(cfg.linear_dim, cfg.input_affine_dim, 1.4)
    sd["encoder.in_linear2.linear.bias"] = (rng.standard_normal(cfg.linear_dim) * 0.1).astype(np.float32)
    for i in range(cfg.fsmn_layers):
        p = f"encoder.fsmn.{i}"
        sd[p + ".linear.linear.weight"] = lin(cfg.proj_dim, cfg.linear_dim, 1.2)
        sd[p + ".fsmn_block.conv_left.weight"] = (rng.standard_normal((cfg.proj_dim, 1, cfg.lorder, 1)) * 0.1).astype(np.float32)
        sd[p + ".affine.linear.weight"] = lin(cfg.linear_dim, cfg.proj_dim, 1.0)
        sd[p + ".affine.linear.bias"] = (rng.standard_normal(cfg.linear_dim) * 0.1).astype(np.float32)
    sd["encoder.out_linear1.linear.weight"] = lin(cfg.output_affine_dim, cfg.linear_dim, 1.2)
    sd["encoder.out_linear1.linear.bias"] = (rng.standard_normal(cfg.output_affine_dim) * 0.1).astype(np.float32)
    sd["encoder.out_linear2.linear.weight"] = lin(cfg.output_dim, cfg.output_affine_dim, 3.0)
    b = (rng.standard_normal(cfg.output_dim) * 0.1).astype(np.float32)
    b[0] += sil_bias
    sd["encoder.out_linear2.linear.bias"] = b
    return sd


def new_cache(cfg: VadNetConfig = VadNetConfig()) -> np.ndarray:
    """The rows in front of a stream's first frame: zeros, [fsmn_layers, lorder - 1, proj_dim]."""
    return np.zeros((cfg.fsmn_layers, cfg.lorder - 1, cfg.proj_dim), dtype=np.float64)


def fsmn_forward(sd: dict, feats: np.ndarray, cache: np.ndarray | None = None, cfg: VadNetConfig = VadNetConfig(),
                 return_logits: bool = False):
    """feats [T, input_dim] -> (scores [T, output_dim] float64, new cache).  ``cache`` as ``new_cache`` gives it / as an
    earlier call returned it (FSMN.forward with ``cache`` of FunASR's encoder: one entry per block)."""
    w = {k: np.asarray(v, dtype=np.float64) for k, v in sd.items()}
    x = np.asarray(feats, dtype=np.float64)
    cache = new_cache(cfg) if cache is None else np.array(cache, dtype=np.float64)
    x = x @ w["encoder.in_linear1.linear.weight"].T + w["encoder.in_linear1.linear.bias"]
    x = x @ w["encoder.in_linear2.linear.weight"].T + w["encoder.in_linear2.linear.bias"]
    x = np.maximum(x, 0.0)
    hist = cfg.lorder - 1
    for i in range(cfg.fsmn_layers):
        pre = f"encoder.fsmn.{i}"
        p = x @ w[pre + ".linear.linear.weight"].T
        taps = w[pre + ".fsmn_block.conv_left.weight"].reshape(cfg.proj_dim, cfg.lorder)
        seq = np.concatenate([cache[i], p], axis=0)          # rows -19 .. T-1
        out = p.copy()
        for j in range(cfg.lorder):
            out += seq[j:j + p.shape[0]] * taps[:, j][None, :]
        cache[i] = seq[seq.shape[0] - hist:]
        x = out @ w[pre + ".affine.linear.weight"].T + w[pre + ".affine.linear.bias"]
        x = np.maximum(x, 0.0)
    x = x @ w["encoder.out_linear1.linear.weight"].T + w["encoder.out_linear1.linear.bias"]
    logits = x @ w["encoder.out_linear2.linear.weight"].T + w["encoder.out_linear2.linear.bias"]
    if return_logits:
        return logits, cache
    z = logits - logits.max(axis=1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=1, keepdims=True), cache


def frame_decibel(pcm_int16: np.ndarray) -> np.ndarray:
    """10 log10(sum x^2 + 1e-6) per 400-sample frame at hop 160 of the waveform in [-1, 1] (E2EVadModel.ComputeDecibel)."""
    x = np.asarray(pcm_int16, dtype=np.float64) / 32768.0
    n = 0 if x.shape[0] < 400 else 1 + (x.shape[0] - 400) // 160
    out = np.empty((n,), dtype=np.float64)
    for f in range(n):
        seg = x[160 * f:160 * f + 400]
        out[f] = 10.0 * np.log10(np.sum(seg * seg) + 1e-6)
    return out


def lfr_offline(logmel: np.ndarray, lfr_m: int = 5) -> np.ndarray:
    """lfr_n = 1: row i = frames i - left .. i + left with the ends repeated (what a whole stream yields once final)."""
    t = logmel.shape[0]
    left = (lfr_m - 1) // 2
    idx = np.clip(np.arange(t)[:, None] + np.arange(-left, lfr_m - left)[None, :], 0, t - 1)
    return logmel[idx].reshape(t, -1)


class OnlineFrontend:
    """WavFrontendOnline restated for one stream: ``push(pcm_int16, is_final)`` -> (feats [n, 400] before CMVN as
    float64, decibels [n]) of the rows that become available with this chunk."""

    def __init__(self, lfr_m: int = 5):
        self.lfr_m = lfr_m
        self.left = (lfr_m - 1) // 2
        self.tail = np.zeros((0,), dtype=np.int16)
        self.splice = None          # the rows kept for the next call (FunASR's lfr_splice_cache)
        self.db_wait = np.zeros((0,), dtype=np.float64)

    def push(self, pcm_int16: np.ndarray, is_final: bool = False):
        x = np.concatenate([self.tail, np.asarray(pcm_int16, dtype=np.int16)])
        n = 0 if x.shape[0] < 400 else 1 + (x.shape[0] - 400) // 160
        empty = (np.zeros((0, 80 * self.lfr_m)), np.zeros((0,)))
        if n == 0:
            self.tail = x
            if not is_final or self.splice is None:
                return empty
            mel = np.zeros((0, 80))
        else:
            mel = fbank_oracle.kaldi_fbank(x[: 160 * (n - 1) + 400])
            self.db_wait = np.concatenate([self.db_wait, frame_decibel(x[: 160 * (n - 1) + 400])])
            self.tail = x[160 * n:]
        if self.splice is None:
            self.splice = np.repeat(mel[:1], self.left, axis=0)
        rows = np.concatenate([self.splice, mel], axis=0)
        t = rows.shape[0]
        out = []
        i = 0
        while i < t - self.left:
            if self.lfr_m <= t - i:
                out.append(rows[i:i + self.lfr_m].reshape(-1))
            elif is_final:
                pad = np.repeat(rows[-1:], self.lfr_m - (t - i), axis=0)
                out.append(np.concatenate([rows[i:], pad], axis=0).reshape(-1))
            else:
                break
            i += 1
        self.splice = rows[min(t - 1, i):] if t else rows
        feats = np.stack(out) if out else np.zeros((0, 80 * self.lfr_m))
        db = self.db_wait[: feats.shape[0]]
        self.db_wait = self.db_wait[feats.shape[0]:]
        return feats, db
