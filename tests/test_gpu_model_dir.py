"""``AutoModel(model=<local directory>, device="cuda:0").generate(input=...)`` on the GPU: the call the reference makes
(voice_interface.py:683-691) on a small seeded model directory.  The results equal those of the engine and the head driven
by hand from the same files; float, int16, tensor and WAV-file inputs of one signal give one result."""
import wave

import numpy as np
import pytest
import torch

from oracle import fbank_oracle as fo
from tests.model_dir_util import write_model_dir

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    import toolbox_for_asr_and_tts_amd as p

    p.load()
    return p


def test_generate_matches_engine_and_head(pkg, tmp_path):
    ref = write_model_dir(str(tmp_path / "paraformer"))
    am = pkg.AutoModel(model=str(tmp_path / "paraformer"), device="cuda:0", batch=4, max_seconds=6.0, disable_update=True,
                       disable_pbar=True)
    clips = [fo.synth_pcm("bursts", 48000, 80), fo.synth_pcm("gauss", 30000, 81), fo.synth_pcm("chirp", 80000, 82),
             fo.synth_pcm("bursts", 16000, 83), fo.synth_pcm("gauss", 64000, 84)]
    clips = [np.asarray(c, dtype=np.int16) for c in clips]
    res = am.generate(input=clips)
    assert [r["key"] for r in res] == [f"rand_key_{i}" for i in range(5)]
    # by hand: the same files through an engine and a head of their own
    from toolbox_for_asr_and_tts_amd import model_dir
    parts = model_dir.read_model_dir(str(tmp_path / "paraformer"))
    eng = pkg.AsrEngine(batch=5, max_samples=96000, cfg=model_dir.library_config(parts["config"]), cmvn=parts["cmvn"],
                        state_dict=parts["state_dict"])
    head = pkg.ParaformerHead(eng.ctx, num_blocks=2, vocab_size=300, token_list=parts["tokens"])
    head.load_state_dict(parts["state_dict"])
    pcm = np.zeros((5, 96000), dtype=np.int16)
    for i, c in enumerate(clips):
        pcm[i, :len(c)] = c
    lens = torch.tensor([len(c) for c in clips], dtype=torch.int32, device="cuda")
    out, olens = eng.run(torch.from_numpy(pcm).cuda(), lens)
    want = head.recognize(out, olens)
    ids = head.forward(out, olens)["token_ids"].cpu().tolist()
    n_tok = 0
    for i in range(5):
        # (the batch of 4 + 1 runs other launch plans than the batch of 5: token ids are compared where the logits
        # are not a near tie, which the seeded weights make rare; the text follows the ids)
        a, b = res[i]["token_ids"], head.clean_ids(ids[i])
        assert len(a) == len(b)
        n_tok += len(a)
        assert sum(x != y for x, y in zip(a, b)) <= max(1, len(a) // 20)
        if a == b:
            assert res[i]["text"] == want[i]
    assert n_tok > 20
    assert ref["tokens"][5] in "".join(ref["tokens"])  # (the list the text was built from)


def test_one_utterance_in_every_input_form(pkg, tmp_path):
    write_model_dir(str(tmp_path / "m"))
    am = pkg.AutoModel(model=str(tmp_path / "m"), device="cuda:0", batch=2, max_seconds=12.0)
    pcm = np.asarray(fo.synth_pcm("bursts", 40000, 90), dtype=np.int16)
    wav = tmp_path / "utt_7.wav"
    with wave.open(str(wav), "wb") as wf:
        wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(16000)
        wf.writeframes(pcm.tobytes())
    r_int = am.generate(input=pcm)
    r_flt = am.generate(input=pcm.astype(np.float32) / 32768.0)
    r_ten = am.generate(input=torch.from_numpy(pcm))
    r_wav = am.generate(input=str(wav))
    assert r_wav[0]["key"] == "utt_7" and r_int[0]["key"] == "rand_key_0"
    assert r_int[0]["token_ids"] == r_flt[0]["token_ids"] == r_ten[0]["token_ids"] == r_wav[0]["token_ids"]
    assert r_int[0]["text"] == r_wav[0]["text"] and len(r_int[0]["token_ids"]) > 3
    assert r_int[0]["sentence_info"] == [{"text": r_int[0]["text"], "start": 0, "end": 2500, "spk": 0}]
    # a single short utterance went through a batch-1 engine sized for 5 s, not through the batch engine
    assert list(am._one) == [5 * 16000] and am._one[80000].tp_max < am.engine.tp_max
    # 8 s: the next size; too long: refused
    am.generate(input=np.asarray(fo.synth_pcm("gauss", 128000, 91), dtype=np.int16))
    assert sorted(am._one) == [80000, 160000]
    with pytest.raises(ValueError, match="exceed"):
        am.generate(input=np.zeros(12 * 16000 + 1, dtype=np.int16))
    assert am.generate(input=[]) == []
