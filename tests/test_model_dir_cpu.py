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


def test_float_input_stays_float(md):
    x = np.array([0.0, 0.25, -1.0, 0.123456], dtype=np.float64)
    got = md.AutoModel._to_pcm(x)
    assert got.dtype == np.float32 and np.array_equal(got, x.astype(np.float32))
    i = np.array([3, -4], dtype=np.int16)
    assert md.AutoModel._to_pcm(i) is i
    assert md.AutoModel._to_pcm(torch.tensor([[0.5], [0.25]])).tolist() == [0.5, 0.25]


def test_decoder_variants_that_are_not_built_are_refused(md, tmp_path):
    ref = write_model_dir(str(tmp_path / "m"))
    cfg, sd = ref["config"], ref["state_dict"]
    md.check_decoder_config(cfg, sd)                                   # the directory as written is fine
    md.check_decoder_config(dict(cfg, predictor="CifPredictorV3"), sd)
    bad = dict(cfg, decoder_conf=dict(cfg["decoder_conf"], att_layer_num=1))
    with pytest.raises(ValueError, match="att_layer_num"):
        md.check_decoder_config(bad, sd)
    with pytest.raises(ValueError, match="decoders2"):
        md.check_decoder_config(cfg, dict(sd, **{"decoder.decoders2.0.norm1.weight": torch.zeros(512)}))
    with pytest.raises(ValueError, match="sanm_shfit"):
        md.check_decoder_config(dict(cfg, decoder_conf=dict(cfg["decoder_conf"], sanm_shfit=5)), sd)
    with pytest.raises(ValueError, match="predictor"):
        md.check_decoder_config(dict(cfg, predictor="CifPredictor"), sd)


def test_token_spans_follow_the_fires():
    from toolbox_for_asr_and_tts_amd.decoder import ParaformerHead
    #        sos  tok  blank tok  eos  pad
    ids = [1, 17, 0, 250, 2, -1]
    fires = [0, 3, 4, 9, 12, -1]
    assert ParaformerHead.clean_ids(ids, 300) == [17, 250]
    assert ParaformerHead.token_spans(ids, fires, 300) == [[60, 240], [300, 600]]
    assert ParaformerHead.clean_ids([5, 0x7fffffff, 299], 300) == [5, 299]    # not a vocabulary index: dropped
