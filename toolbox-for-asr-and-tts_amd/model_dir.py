"""Loading a local Paraformer model directory, as the reference does with ``funasr.AutoModel(model=<dir>, device="cuda:0",
disable_update=True, disable_pbar=True)`` (/root/reference/voice-service/app/services/voice_interface.py:683-691), and
the ``generate(input=...)`` call that returns ``[{"key": ..., "text": ...}]``.

A model directory of the ModelScope / FunASR layout holds
    config.yaml   frontend_conf / encoder_conf / predictor_conf / decoder_conf (and optionally ``token_list``)
    am.mvn        CMVN statistics (Kaldi nnet text: <AddShift>, <Rescale>)
    model.pt      the checkpoint: a ``state_dict`` (possibly under the key "state_dict" or "model")
(an FSMN-VAD directory holds ``vad.mvn`` instead of ``am.mvn`` and no token list)
    tokens.json   the token list (or tokens.txt, one token per line)
Nothing is executed from those files: YAML is read with the safe loader, the checkpoint with
``torch.load(weights_only=True)``.  Everything runs on the device through libasrfe; there is no CPU path."""
from __future__ import annotations

import json
import os
import wave

import numpy as np
import torch
import yaml

from . import _lib
from .decoder import ParaformerHead
from .engine import AsrEngine
from .frontend import load_cmvn


def read_model_dir(path: str) -> dict:
    """-> {"config": dict, "cmvn": (shift, scale), "tokens": list | None, "state_dict": dict}; raises FileNotFoundError /
    ValueError with the offending file's name."""
    if not os.path.isdir(path):
        raise FileNotFoundError(f"{path}: not a model directory")
    cfg_file = os.path.join(path, "config.yaml")
    if not os.path.isfile(cfg_file):
        raise FileNotFoundError(f"{cfg_file}: missing")
    with open(cfg_file, "r", encoding="utf-8") as f:
        config = yaml.safe_load(f) or {}
    if not isinstance(config, dict):
        raise ValueError(f"{cfg_file}: not a mapping")
    cmvn_file = os.path.join(path, str((config.get("frontend_conf") or {}).get("cmvn_file") or "am.mvn"))
    for name in ("am.mvn", "vad.mvn"):
        if not os.path.isfile(cmvn_file):
            cmvn_file = os.path.join(path, name)
    cmvn = load_cmvn(cmvn_file)
    tokens = None
    for name in ("tokens.json", "tokens.txt"):
        tf = os.path.join(path, name)
        if os.path.isfile(tf):
            with open(tf, "r", encoding="utf-8") as f:
                tokens = json.load(f) if name.endswith(".json") else [ln.rstrip("\n").split(" ")[0] for ln in f if ln.strip()]
            break
    if tokens is None and isinstance(config.get("token_list"), list):
        tokens = list(config["token_list"])
    ckpt = None
    for name in ("model.pt", "model.pb", "model.pth"):
        cf = os.path.join(path, name)
        if os.path.isfile(cf):
            ckpt = torch.load(cf, map_location="cpu", weights_only=True)
            break
    if ckpt is None:
        raise FileNotFoundError(f"{path}: no model.pt")
    for key in ("state_dict", "model"):
        if isinstance(ckpt, dict) and isinstance(ckpt.get(key), dict):
            ckpt = ckpt[key]
    if not isinstance(ckpt, dict) or not any(k.startswith("encoder.") for k in ckpt):
        raise ValueError(f"{path}: the checkpoint holds no encoder.* tensors")
    return {"config": config, "cmvn": cmvn, "tokens": tokens, "state_dict": ckpt}


def is_vad_config(config: dict) -> bool:
    """True for a config.yaml of FunASR's streaming FSMN voice-activity model."""
    return str(config.get("model", "")).startswith("FsmnVAD") or str(config.get("encoder", "")) == "FSMN"


def library_config(config: dict, small_batch_rows: int | None = None) -> _lib.Config:
    """The library configuration a FunASR ``config.yaml`` asks for (missing entries keep paraformer-large's values)."""
    c = _lib.default_config()
    fe = config.get("frontend_conf") or {}
    en = config.get("encoder_conf") or {}
    c.sample_rate = int(fe.get("fs", c.sample_rate))
    c.n_mels = int(fe.get("n_mels", c.n_mels))
    c.frame_length = int(round(float(fe.get("frame_length", 25)) * c.sample_rate / 1000.0))
    c.frame_shift = int(round(float(fe.get("frame_shift", 10)) * c.sample_rate / 1000.0))
    c.lfr_m = int(fe.get("lfr_m", c.lfr_m))
    c.lfr_n = int(fe.get("lfr_n", c.lfr_n))
    window = str(fe.get("window", "hamming"))
    if window != "hamming":
        raise ValueError(f"frontend_conf.window = {window!r}: only the Hamming window is built")
    # (a dither entry is ignored: FunASR's own inference path runs the frontend with dither switched off as well)
    c.enc_input_size = c.lfr_m * c.n_mels
    c.enc_output_size = int(en.get("output_size", c.enc_output_size))
    c.enc_heads = int(en.get("attention_heads", c.enc_heads))
    c.enc_linear_units = int(en.get("linear_units", c.enc_linear_units))
    c.enc_num_blocks = int(en.get("num_blocks", c.enc_num_blocks))
    c.enc_kernel_size = int(en.get("kernel_size", c.enc_kernel_size))
    c.enc_sanm_shift = int(en.get("sanm_shfit", en.get("sanm_shift", c.enc_sanm_shift)))
    if small_batch_rows is not None:
        c.enc_small_batch_rows = int(small_batch_rows)
    return c


def check_decoder_config(config: dict, state_dict: dict) -> None:
    """Raises ValueError for a Paraformer directory whose predictor / decoder differs from the ones built here (its extra
    tensors would otherwise be ignored without a word and the text would be wrong)."""
    de = config.get("decoder_conf") or {}
    blocks = int(de.get("num_blocks", 16))
    att = int(de.get("att_layer_num", blocks))
    if att != blocks:
        raise ValueError(f"decoder_conf.att_layer_num = {att} with num_blocks = {blocks}: blocks without cross-attention are not built")
    if any(k.startswith("decoder.decoders2.") for k in state_dict):
        raise ValueError("the checkpoint holds decoder.decoders2.* (blocks without cross-attention): not built")
    shift = int(de.get("sanm_shfit", de.get("sanm_shift", 0)))
    if shift != 0:
        raise ValueError(f"decoder_conf.sanm_shfit = {shift}: only the symmetric FSMN memory block is built")
    predictor = config.get("predictor")
    if predictor is not None and str(predictor) not in ("CifPredictorV2", "CifPredictorV3"):
        raise ValueError(f"predictor = {predictor!r}: only CifPredictorV2 / CifPredictorV3 are built")


def _read_wav(path: str) -> np.ndarray:
    with wave.open(path, "rb") as wf:
        if wf.getsampwidth() != 2:
            raise ValueError(f"{path}: only 16-bit PCM WAV files are read here")
        if wf.getframerate() != 16000:
            raise ValueError(f"{path}: {wf.getframerate()} Hz; the model takes 16 kHz")
        pcm = np.frombuffer(wf.readframes(wf.getnframes()), dtype=np.int16)
        if wf.getnchannels() > 1:
            pcm = pcm.reshape(-1, wf.getnchannels()).astype(np.int32).mean(axis=1).astype(np.int16)
    return pcm


class AutoModel:
    """``AutoModel(model=<local directory>, device="cuda:0")`` for a Paraformer directory; ``generate(input=...)`` takes
    what the reference passes (a WAV path, a float32 numpy array in [-1, 1] at 16 kHz, an int16 array, a torch tensor, or
    a list of those) and returns one ``{"key", "text", "timestamp", "sentence_info"}`` per utterance (``"token_ids"``
    too; text needs the directory's token list).  ``timestamp`` holds [start_ms, end_ms] per token from the frames at
    which the CIF predictor fired (60 ms frames); ``sentence_info`` is the whole utterance as one sentence of speaker 0
    from the first token's start to the last token's end (sentence splitting and speaker labels come from the
    punctuation and speaker models, which are not part of this package; a BiCif checkpoint's own time-stamp head,
    ``CifPredictorV3``'s upsampling branch, is not used either).  Float input goes to the device as float (the frontend
    scales it by 32768, as FunASR's does); int16 input as int16.  ``batch`` utterances of up to ``max_seconds`` go through
    the device at once; longer inputs are refused (the reference's VAD cuts them into segments first).

    A directory whose decoder this package does not build is refused at load: ``att_layer_num`` other than
    ``num_blocks`` (blocks without cross-attention, ``decoder.decoders2.*``), a decoder ``sanm_shfit`` other than 0
    (asymmetric FSMN padding), or a predictor other than ``CifPredictorV2`` / ``CifPredictorV3``."""

    def __init__(self, model: str, device: str = "cuda:0", batch: int = 16, max_seconds: float = 30.0,
                 disable_update: bool = True, disable_pbar: bool = True, small_batch_rows: int | None = None, **kwargs):
        if not str(device).startswith("cuda"):
            raise ValueError("AutoModel: this package runs on a GPU only (device='cuda:N')")
        parts = read_model_dir(model)
        self.model_path = model
        self.device = torch.device(device)
        self.kwargs = parts["config"]
        self._vad = None
        if is_vad_config(parts["config"]):
            # the streaming voice-activity model (voice_interface.py:723-727): its generate() takes the per-chunk
            # arguments the reference passes at :1585-1590
            from .vad import FsmnVadModel
            self._vad = FsmnVadModel(parts["config"], parts["cmvn"], parts["state_dict"], device=self.device)
            return
        cfg = library_config(parts["config"], small_batch_rows)
        self.max_samples = int(max_seconds * cfg.sample_rate)
        sd = parts["state_dict"]
        self.engine = AsrEngine(batch=batch, max_samples=self.max_samples, cfg=cfg, device=self.device, cmvn=parts["cmvn"],
                                state_dict=sd)
        # one utterance at a time (the service's shape) goes through engines of batch 1 on the same context and weights,
        # sized for 5 / 10 / 20 s and the full length and made on first use: a short call then computes its own frames
        # only and takes the library's latency plan instead of a batch of padding
        self._one = {}
        de = parts["config"].get("decoder_conf") or {}
        pr = parts["config"].get("predictor_conf") or {}
        check_decoder_config(parts["config"], sd)
        vocab = len(parts["tokens"]) if parts["tokens"] else int(sd["decoder.output_layer.weight"].shape[0])
        self.head = ParaformerHead(self.engine.ctx, num_blocks=int(de.get("num_blocks", 16)),
                                   linear_units=int(de.get("linear_units", 2048)), vocab_size=vocab,
                                   kernel_size=int(de.get("kernel_size", 11)),
                                   tail_threshold=float(pr.get("tail_threshold", 0.45)), threshold=float(pr.get("threshold", 1.0)),
                                   smooth_factor=float(pr.get("smooth_factor", 1.0)),
                                   noise_threshold=float(pr.get("noise_threshold", 0.0)), token_list=parts["tokens"])
        self.head.load_state_dict(sd)
        self.batch = batch

    def _engine_for(self, n_clips: int, n_samples: int) -> AsrEngine:
        if n_clips > 1:
            return self.engine
        sr = self.engine.ctx.cfg.sample_rate
        for seconds in (5, 10, 20):
            if n_samples <= seconds * sr < self.max_samples:
                key = seconds * sr
                break
        else:
            key = self.max_samples
        if key not in self._one:
            self._one[key] = AsrEngine(batch=1, max_samples=key, device=self.device, ctx=self.engine.ctx)
        return self._one[key]

    # ---- input handling ----------------------------------------------------------------------------------------------
    @staticmethod
    def _to_pcm(x) -> np.ndarray:
        """-> int16 samples (a WAV path, an int16 array) or float32 samples in [-1, 1) (anything else).  Float samples stay
        float: FunASR's frontend multiplies them by 32768 without rounding to the int16 grid, and so does the device
        frontend; a resampled chunk is then not quantised a second time."""
        if isinstance(x, str):
            return _read_wav(x)
        if isinstance(x, torch.Tensor):
            x = x.detach().cpu().numpy()
        a = np.asarray(x)
        if a.ndim != 1:
            a = a.reshape(-1)
        if a.dtype == np.int16:
            return a
        return np.ascontiguousarray(a, dtype=np.float32)

    @staticmethod
    def _to_int16(x) -> np.ndarray:
        """int16 samples of any input (the streaming voice-activity frontend keeps its sample tail as int16): float PCM
        is rounded to the int16 grid, which is what a WAV file of the same signal would hold."""
        a = AutoModel._to_pcm(x)
        if a.dtype == np.int16:
            return a
        return np.clip(np.rint(a.astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16)

    def generate(self, input, cache=None, is_final: bool | None = None, **kwargs) -> list:
        if self._vad is not None:
            return self._vad.generate(input, cache=cache, is_final=is_final, **kwargs)
        items = list(input) if isinstance(input, (list, tuple)) else [input]
        keys = [os.path.splitext(os.path.basename(x))[0] if isinstance(x, str) else f"rand_key_{i}" for i, x in enumerate(items)]
        clips = [self._to_pcm(x) for x in items]
        for k, c in zip(keys, clips):
            if c.shape[0] > self.max_samples:
                raise ValueError(f"{k}: {c.shape[0]} samples exceed the engine's {self.max_samples}; cut the input into segments")
        results = [None] * len(clips)
        order = sorted(range(len(clips)), key=lambda i: clips[i].shape[0])  # length-sorted batches, as FunASR builds them
        for lo in range(0, len(order), self.batch):
            idx = order[lo:lo + self.batch]
            n = max(clips[i].shape[0] for i in idx)
            eng = self._engine_for(len(idx), n)
            # a batch with any float clip travels as float32 (int16 / 32768 is exact in float32 and comes back as the
            # same integer after the frontend's scaling, so the int16 clips beside it are not changed by that)
            as_float = any(clips[i].dtype != np.int16 for i in idx)
            pcm = np.zeros((eng.batch, max(n, 1)), dtype=np.float32 if as_float else np.int16)
            lens = np.zeros((eng.batch,), dtype=np.int32)
            for row, i in enumerate(idx):
                c = clips[i]
                pcm[row, :c.shape[0]] = c.astype(np.float32) / np.float32(32768.0) if (as_float and c.dtype == np.int16) else c
                lens[row] = c.shape[0]
            out, olens = eng.run(torch.from_numpy(pcm).to(self.device), torch.from_numpy(lens).to(self.device))
            r = self.head.forward(out, olens, return_fire_frames=True)
            rows = r["token_ids"].cpu().tolist()
            fires = r["fire_frames"].cpu().tolist()
            for row, i in enumerate(idx):
                ids = self.head.clean_ids(rows[row], self.head.vocab_size)
                item = {"key": keys[i], "token_ids": ids}
                item["text"] = self.head.ids2text(ids) if self.head.token_list is not None else ""
                # the reference reads rec_result[0]["sentence_info"] = [{"text", "start" (ms), "end" (ms), "spk"}]
                # (voice_interface.py:2064-2067, 2092-2093, 2124-2126).  Times come from the frames at which the
                # predictor fired (a token ends with the frame that completes it); sentence splitting and speaker
                # labels come from models that are not part of this package: one sentence of speaker 0
                dur_ms = int(round(1000.0 * clips[i].shape[0] / self.engine.ctx.cfg.sample_rate))
                spans = [[min(a, dur_ms), min(b, dur_ms)] for a, b in self.head.token_spans(rows[row], fires[row], self.head.vocab_size)]
                item["timestamp"] = spans
                item["sentence_info"] = [{"text": item["text"], "start": spans[0][0], "end": spans[-1][1], "spk": 0,
                                          "timestamp": spans}] if ids else []
                results[i] = item
        return results
