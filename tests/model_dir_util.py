"""Writes a small model directory of the ModelScope / FunASR layout (config.yaml, am.mvn, tokens.json, model.pt) from
seeded weights: what the model-directory tests load."""
import json
import os

import numpy as np
import torch
import yaml

from oracle import paraformer_decoder_oracle as po
from oracle import sanm_oracle as so

VOCAB = 300


def token_list():
    toks = ["<blank>", "<s>", "</s>"] + [chr(0x4E00 + i) for i in range(200)] + [f"w{i}@@" for i in range(40)] + [f"en{i}" for i in range(56)]
    toks.append("<unk>")
    assert len(toks) == VOCAB
    return toks


def write_model_dir(path, enc_blocks=2, dec_blocks=2, seed=21, wrap_key=None, tokens_as="json"):
    os.makedirs(path, exist_ok=True)
    config = {
        "encoder": "SANMEncoder",
        "encoder_conf": {"output_size": 512, "attention_heads": 4, "linear_units": 2048, "num_blocks": enc_blocks,
                         "kernel_size": 11, "sanm_shfit": 0},
        "decoder": "ParaformerSANMDecoder",
        "decoder_conf": {"attention_heads": 4, "linear_units": 2048, "num_blocks": dec_blocks, "kernel_size": 11},
        "predictor": "CifPredictorV2",
        "predictor_conf": {"idim": 512, "threshold": 1.0, "l_order": 1, "r_order": 1, "tail_threshold": 0.45},
        "frontend": "WavFrontend",
        "frontend_conf": {"fs": 16000, "window": "hamming", "n_mels": 80, "frame_length": 25, "frame_shift": 10, "lfr_m": 7,
                          "lfr_n": 6, "cmvn_file": "am.mvn"},
    }
    with open(os.path.join(path, "config.yaml"), "w", encoding="utf-8") as f:
        yaml.safe_dump(config, f)
    rng = np.random.default_rng(seed)
    shift = (-8.0 + rng.standard_normal(560)).astype(np.float32)
    scale = (0.2 + 0.05 * rng.random(560)).astype(np.float32)
    with open(os.path.join(path, "am.mvn"), "w", encoding="utf-8") as f:
        f.write("<Nnet>\n<Splice> 560 560\n[ 0 ]\n<AddShift> 560 560\n")
        f.write("<LearnRateCoef> 0 [ " + " ".join(repr(float(v)) for v in shift) + " ]\n")
        f.write("<Rescale> 560 560\n")
        f.write("<LearnRateCoef> 0 [ " + " ".join(repr(float(v)) for v in scale) + " ]\n</Nnet>\n")
    toks = token_list()
    if tokens_as == "json":
        with open(os.path.join(path, "tokens.json"), "w", encoding="utf-8") as f:
            json.dump(toks, f, ensure_ascii=False)
    elif tokens_as == "txt":
        with open(os.path.join(path, "tokens.txt"), "w", encoding="utf-8") as f:
            f.write("\n".join(toks) + "\n")
    sd = {"encoder." + k: v for k, v in so.make_state_dict(so.SanmConfig(num_blocks=enc_blocks), seed=seed).items()}
    sd.update(po.make_state_dict(po.DecoderConfig(num_blocks=dec_blocks, vocab_size=VOCAB), seed=seed + 1))
    sd = {k: v.float() for k, v in sd.items()}
    torch.save({wrap_key: sd} if wrap_key else sd, os.path.join(path, "model.pt"))
    return {"config": config, "shift": shift, "scale": scale, "tokens": toks, "state_dict": sd}
