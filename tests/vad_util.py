"""Helpers of the voice-activity tests: a hand-built "energy" network under FunASR's tensor names (its silence
probability follows the level of the audio, so that bursts of noise come out as segments), test signals, and a writer for a
model directory of the FSMN-VAD layout (config.yaml, vad.mvn, model.pt)."""
import os

import numpy as np
import torch
import yaml

from oracle import fsmn_vad_oracle as vo

CMVN_SHIFT, CMVN_SCALE = -10.0, 0.2


def energy_state_dict(cfg: vo.VadNetConfig = vo.VadNetConfig(), seed: int = 5, tap_scale: float = 0.0) -> dict:
    """Channel 0 carries relu(E), channel 1 relu(-E) with E = the mean of the 400 normalised log-mel inputs; the silence
    logit is 3 (c1 - c0), class 1 the opposite, every other class far below.  Small seeded values everywhere else keep the
    other paths of the kernels busy without moving the decision."""
    rng = np.random.default_rng(seed)
    sd = {}

    def small(*shape):
        return (rng.standard_normal(shape) * 1e-3).astype(np.float32)

    w = small(cfg.input_affine_dim, cfg.input_dim); w[0, :] = 1.0 / cfg.input_dim
    sd["encoder.in_linear1.linear.weight"] = w
    sd["encoder.in_linear1.linear.bias"] = np.zeros(cfg.input_affine_dim, np.float32)
    w = small(cfg.linear_dim, cfg.input_affine_dim); w[0, 0] = 1.0; w[1, 0] = -1.0
    sd["encoder.in_linear2.linear.weight"] = w
    sd["encoder.in_linear2.linear.bias"] = np.zeros(cfg.linear_dim, np.float32)
    for i in range(cfg.fsmn_layers):
        p = f"encoder.fsmn.{i}"
        w = small(cfg.proj_dim, cfg.linear_dim); w[0, 0] = 1.0; w[1, 1] = 1.0
        sd[p + ".linear.linear.weight"] = w
        sd[p + ".fsmn_block.conv_left.weight"] = (rng.standard_normal((cfg.proj_dim, 1, cfg.lorder, 1)) * tap_scale).astype(np.float32)
        w = small(cfg.linear_dim, cfg.proj_dim); w[0, 0] = 1.0; w[1, 1] = 1.0
        sd[p + ".affine.linear.weight"] = w
        sd[p + ".affine.linear.bias"] = np.zeros(cfg.linear_dim, np.float32)
    w = small(cfg.output_affine_dim, cfg.linear_dim); w[0, 0] = 1.0; w[1, 1] = 1.0
    sd["encoder.out_linear1.linear.weight"] = w
    sd["encoder.out_linear1.linear.bias"] = np.zeros(cfg.output_affine_dim, np.float32)
    w = small(cfg.output_dim, cfg.output_affine_dim); w[0, 0] = -3.0; w[0, 1] = 3.0; w[1, 0] = 3.0; w[1, 1] = -3.0
    sd["encoder.out_linear2.linear.weight"] = w
    b = np.full(cfg.output_dim, -20.0, np.float32); b[0] = 0.0; b[1] = 0.0
    sd["encoder.out_linear2.linear.bias"] = b
    return sd


def cmvn(dim: int = 400):
    return np.full(dim, CMVN_SHIFT, np.float32), np.full(dim, CMVN_SCALE, np.float32)


def bursts(n_samples: int, spans, seed: int = 0, loud: float = 3000.0, quiet: float = 3.0) -> np.ndarray:
    """int16 noise: ``quiet`` everywhere, ``loud`` inside the spans (in samples)."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n_samples) * quiet
    for a, b in spans:
        x[a:b] = rng.standard_normal(b - a) * loud
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def vad_config() -> dict:
    return {
        "model": "FsmnVADStreaming",
        "model_conf": {"sample_rate": 16000, "detect_mode": 1, "max_end_silence_time": 800, "max_start_silence_time": 3000,
                       "window_size_ms": 200, "sil_to_speech_time_thres": 150, "speech_to_sil_time_thres": 150,
                       "speech_2_noise_ratio": 1.0, "do_extend": 1, "lookback_time_start_point": 200,
                       "lookahead_time_end_point": 100, "max_single_segment_time": 60000, "snr_thres": -100.0,
                       "noise_frame_num_used_for_snr": 100, "decibel_thres": -100.0, "speech_noise_thres": 0.6,
                       "fe_prior_thres": 0.0001, "silence_pdf_num": 1, "sil_pdf_ids": [0], "frame_in_ms": 10,
                       "frame_length_ms": 25},
        "encoder": "FSMN",
        "encoder_conf": {"input_dim": 400, "input_affine_dim": 140, "fsmn_layers": 4, "linear_dim": 250, "proj_dim": 128,
                         "lorder": 20, "rorder": 0, "lstride": 1, "rstride": 0, "output_affine_dim": 140, "output_dim": 248},
        "frontend": "WavFrontendOnline",
        "frontend_conf": {"fs": 16000, "window": "hamming", "n_mels": 80, "frame_length": 25, "frame_shift": 10,
                          "dither": 0.0, "lfr_m": 5, "lfr_n": 1},
    }


def write_vad_dir(path, sd=None):
    os.makedirs(path, exist_ok=True)
    config = vad_config()
    with open(os.path.join(path, "config.yaml"), "w", encoding="utf-8") as f:
        yaml.safe_dump(config, f)
    shift, scale = cmvn()
    with open(os.path.join(path, "vad.mvn"), "w", encoding="utf-8") as f:
        f.write("<Nnet>\n<Splice> 400 400\n[ 0 ]\n<AddShift> 400 400\n")
        f.write("<LearnRateCoef> 0 [ " + " ".join(repr(float(v)) for v in shift) + " ]\n")
        f.write("<Rescale> 400 400\n")
        f.write("<LearnRateCoef> 0 [ " + " ".join(repr(float(v)) for v in scale) + " ]\n</Nnet>\n")
    sd = energy_state_dict() if sd is None else sd
    torch.save({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, os.path.join(path, "model.pt"))
    return {"config": config, "shift": shift, "scale": scale, "state_dict": sd}


def oracle_sil_trace(pcm: np.ndarray, sd: dict, chunk: int | None = None):
    """The oracle's silence probability and level per frame of one clip, in one piece or pushed in chunks of ``chunk``
    samples (streaming frontend + cached network)."""
    fe = vo.OnlineFrontend()
    shift, scale = cmvn()
    cache = vo.new_cache()
    sil, db = [], []
    pieces = [pcm] if chunk is None else [pcm[i:i + chunk] for i in range(0, len(pcm), chunk)]
    for i, piece in enumerate(pieces):
        feats, d = fe.push(piece, i == len(pieces) - 1)
        if feats.shape[0]:
            scores, cache = vo.fsmn_forward(sd, (feats + shift) * scale, cache)
            sil.append(scores[:, 0])
            db.append(d)
    return np.concatenate(sil), np.concatenate(db)
