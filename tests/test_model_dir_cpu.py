"""Reading a model directory of the ModelScope / FunASR layout (the argument of the reference's
``AutoModel(model=<local path>)``, voice_interface.py:683-691): files, configuration mapping, refusals.  No GPU."""
import os
import wave

import numpy as np
import pytest
import torch

from tests.model_dir_util import VOCAB, write_model_dir


@pytest.fixture(scope="module")
def md():
    from toolbox_for_asr_and_tts_amd import model_dir
    return model_dir


def test_reads_every_part(md, tmp_path):
    ref = write_model_dir(str(tmp_path / "m"), enc_blocks=3, dec_blocks=2)
    got = md.read_model_dir(str(tmp_path / "m"))
    assert got["config"]["encoder_conf"]["num_blocks"] == 3
    assert np.array_equal(got["cmvn"][0], ref["shift"]) and np.array_equal(got["cmvn"][1], ref["scale"])
    assert got["tokens"] == ref["tokens"] and len(got["tokens"]) == VOCAB
    assert set(got["state_dict"]) == set(ref["state_dict"])
    k = "encoder.encoders.1.self_attn.linear_q_k_v.weight"
    assert torch.equal(got["state_dict"][k], ref["state_dict"][k])


@pytest.mark.parametrize("wrap,tokens_as", [("state_dict", "txt"), ("model", "none")])
def test_wrapped_checkpoint_and_token_sources(md, tmp_path, wrap, tokens_as):
    ref = write_model_dir(str(tmp_path / "m"), wrap_key=wrap, tokens_as=tokens_as)
    got = md.read_model_dir(str(tmp_path / "m"))
    assert set(got["state_dict"]) == set(ref["state_dict"])
    assert got["tokens"] == (ref["tokens"] if tokens_as == "txt" else None)


def test_library_config_follows_the_yaml(md):
    c = md.library_config({"frontend_conf": {"n_mels": 64, "lfr_m": 5, "lfr_n": 1, "frame_length": 25, "frame_shift": 10},
                           "encoder_conf": {"num_blocks": 7, "linear_units": 1024, "sanm_shfit": 2}}, small_batch_rows=0)
    assert (c.n_mels, c.lfr_m, c.lfr_n, c.frame_length, c.frame_shift) == (64, 5, 1, 400, 160)
    assert (c.enc_input_size, c.enc_num_blocks, c.enc_linear_units, c.enc_sanm_shift, c.enc_small_batch_rows) == (320, 7, 1024, 2, 0)
    d = md.library_config({})
    assert (d.n_mels, d.lfr_m, d.lfr_n, d.enc_num_blocks, d.enc_linear_units) == (80, 7, 6, 50, 2048)
    with pytest.raises(ValueError, match="Hamming"):
        md.library_config({"frontend_conf": {"window": "povey"}})


def test_refusals(md, tmp_path):
    with pytest.raises(FileNotFoundError):
        md.read_model_dir(str(tmp_path / "nothing"))
    p = tmp_path / "m"
    write_model_dir(str(p))
    os.remove(p / "model.pt")
    with pytest.raises(FileNotFoundError, match="model.pt"):
        md.read_model_dir(str(p))
    torch.save({"decoder.x": torch.zeros(1)}, p / "model.pt")
    with pytest.raises(ValueError, match="encoder"):
        md.read_model_dir(str(p))
    os.remove(p / "am.mvn")
    with pytest.raises(FileNotFoundError):
        md.read_model_dir(str(p))
    with pytest.raises(ValueError, match="GPU only"):
        md.AutoModel(model=str(p), device="cpu")


def test_input_conversion(md, tmp_path):
    x = np.array([0.0, 0.5, -1.0, 0.99999, 1.5], dtype=np.float32)
    assert md.AutoModel._to_int16(x).tolist() == [0, 16384, -32768, 32767, 32767]
    assert md.AutoModel._to_int16(torch.tensor([1, -2, 3], dtype=torch.int16)).tolist() == [1, -2, 3]
    wav = tmp_path / "a.wav"
    pcm = (np.arange(320) - 160).astype(np.int16)
    with wave.open(str(wav), "wb") as wf:
        wf.setnchannels(2); wf.setsampwidth(2); wf.setframerate(16000)
        wf.writeframes(np.stack([pcm, pcm], axis=1).tobytes())
    assert np.array_equal(md.AutoModel._to_int16(str(wav)), pcm)
    with wave.open(str(wav), "wb") as wf:
        wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(8000)
        wf.writeframes(pcm.tobytes())
    with pytest.raises(ValueError, match="16 kHz"):
        md.AutoModel._to_int16(str(wav))
