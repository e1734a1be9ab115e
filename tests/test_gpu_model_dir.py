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
    fw = head.forward(out, olens, return_logits=True, return_fire_frames=True)
    ids = fw["token_ids"].cpu().tolist()
    n_tok = n_sure = 0
    for i in range(5):
        # (the batch of 4 + 1 runs other launch plans than the batch of 5, whose sums round differently in the last bits:
        # every token whose best logit leads the runner-up by more than 1e-2 must be the same token, and where all are,
        # the text and the time stamps are the same too)
        n = int(fw["token_num"][i])
        top2 = fw["logits"][i, :n].topk(2, dim=-1).values
        sure = ((top2[:, 0] - top2[:, 1]) > 1e-2).cpu().tolist()
        a, b = res[i]["token_ids"], head.clean_ids(ids[i], 300)
        assert len(a) == len(b)
        kept = [s for s, t in zip(sure, ids[i][:n]) if t not in (0, 1, 2)]
        assert all(x == y for x, y, s in zip(a, b, kept) if s)
        n_tok += len(a)
        n_sure += sum(kept)
        if a == b:
            assert res[i]["text"] == want[i]
            dur = round(1000 * len(clips[i]) / 16000)   # (the tail weight may fire one frame behind the last sample)
            spans = head.token_spans(ids[i], fw["fire_frames"][i].cpu().tolist(), 300)
            assert res[i]["timestamp"] == [[min(x, dur), min(y, dur)] for x, y in spans]
    assert n_tok > 20 and n_sure > n_tok // 2
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
    spans = r_int[0]["timestamp"]
    assert len(spans) == len(r_int[0]["token_ids"]) and spans == r_wav[0]["timestamp"] == r_flt[0]["timestamp"]
    assert all(0 <= a < b <= 2500 for a, b in spans) and all(x[1] <= y[0] for x, y in zip(spans, spans[1:]))
    assert all(a % 60 == 0 for a, _ in spans)
    assert r_int[0]["sentence_info"] == [{"text": r_int[0]["text"], "start": spans[0][0], "end": spans[-1][1], "spk": 0,
                                          "timestamp": spans}]
    # float samples that are not on the int16 grid stay float on their way to the device: the result is that of the
    # frontend fed with the float tensor, not that of the rounded samples
    off_grid = (pcm.astype(np.float32) + 0.37) / 32768.0
    r_off = am.generate(input=off_grid)
    eng1 = am._one[5 * 16000]
    out_f, ol_f = eng1.run(torch.from_numpy(off_grid[None]).cuda(), torch.tensor([off_grid.size], dtype=torch.int32, device="cuda"))
    assert r_off[0]["token_ids"] == am.head.clean_ids(am.head.forward(out_f, ol_f)["token_ids"][0].cpu().tolist(), 300)
    # a single short utterance went through a batch-1 engine sized for 5 s, not through the batch engine
    assert list(am._one) == [5 * 16000] and am._one[80000].tp_max < am.engine.tp_max
    # 8 s: the next size; too long: refused
    am.generate(input=np.asarray(fo.synth_pcm("gauss", 128000, 91), dtype=np.int16))
    assert sorted(am._one) == [80000, 160000]
    with pytest.raises(ValueError, match="exceed"):
        am.generate(input=np.zeros(12 * 16000 + 1, dtype=np.int16))
    assert am.generate(input=[]) == []
