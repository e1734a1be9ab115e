"""CPU checks of the voice-activity path: the detector's state machine against segment boundaries derived by hand from its
rules, and the oracle's streaming pieces (cached memory blocks, online frontend) against their one-shot forms."""
import numpy as np
import pytest

from oracle import fbank_oracle as fo
from oracle import fsmn_vad_oracle as vo
from tests import vad_util as vu
from toolbox_for_asr_and_tts_amd.vad import VadDetector, VadOptions, WindowDetector


def sil_trace(n, spans):
    s = np.ones(n)
    for a, b in spans:
        s[a:b] = 0.0
    return s


def level(n):
    return np.full(n, -20.0)


def test_window_detector_thresholds():
    w = WindowDetector(200, 150, 150, 10)
    assert w.get_win_size() == 20
    changes = [w.detect_one_frame(1) for _ in range(20)]
    assert changes[:14] == [2] * 14 and changes[14] == 3 and changes[15:] == [0] * 5   # the 15th speech frame turns it
    changes = [w.detect_one_frame(0) for _ in range(30)]
    # the window falls 19, 18, ... and turns back at 15 speech frames of 20: the fifth silent frame
    assert changes[:4] == [0] * 4 and changes[4] == 1 and all(c == 2 for c in changes[5:])


def test_one_segment_boundaries_follow_the_rules():
    # speech in frames 100..299 of 500.  The window (20 frames, 15 needed) turns at frame 114; the start goes back by the
    # window plus the 200 ms look-back: 114 - 40 = 74 -> 740 ms.  After frame 299 the window falls to 15 at frame 304;
    # 650 ms (max_end_silence 800 - speech_to_sil 150) of continued silence end at frame 369; the end goes back by
    # 65 - 10 (look-ahead) - 1 = 54 frames: frame 315, reported as its end: 3160 ms
    d = VadDetector()
    assert d.step(sil_trace(500, [(100, 300)]), level(500), True, False) == [[740, 3160]]


def test_streaming_reports_open_then_close():
    s = sil_trace(500, [(100, 300)])
    d = VadDetector()
    got = []
    for i in range(0, 500, 40):
        got.append(d.step(s[i:i + 40], level(len(s[i:i + 40])), i + 40 >= 500, True))
    flat = [(i, seg) for i, segs in enumerate(got) for seg in segs]
    assert flat == [(2, [740, -1]), (9, [-1, 3160])]              # frame 114 is in chunk 2, frame 369 in chunk 9


def test_two_segments_no_speech_and_speech_to_the_end():
    d = VadDetector()
    assert d.step(sil_trace(1000, [(100, 300), (450, 700)]), level(1000), True, False) == [[740, 3160], [4240, 7160]]
    assert d.step(sil_trace(300, []), level(300), True, False) == []
    # the stream ends inside speech: the segment closes at the last frame
    assert d.step(sil_trace(300, [(100, 300)]), level(300), True, False) == [[740, 3000]]


def test_a_segment_is_cut_at_the_maximum_length():
    d = VadDetector()
    got = d.step(sil_trace(7200, [(100, 7000)]), level(7200), True, False)
    assert got == [[740, 60750], [60750, 70160]]                    # 6000 frames after frame 74, then a new segment at once


def test_chunking_never_changes_the_segments():
    rng = np.random.default_rng(3)
    s = np.clip(sil_trace(3000, [(200, 650), (900, 1000), (1500, 2600)]) + rng.normal(0, 0.05, 3000), 0, 1)
    whole = VadDetector().step(s, level(3000), True, False)
    assert len(whole) == 3
    for chunk in (1, 7, 40, 333):
        d = VadDetector()
        segs = []
        for i in range(0, 3000, chunk):
            segs += d.step(s[i:i + chunk], level(len(s[i:i + chunk])), i + chunk >= 3000, False)
        assert segs == whole
        d = VadDetector()
        opened, closed = [], []
        for i in range(0, 3000, chunk):
            for b, e in d.step(s[i:i + chunk], level(len(s[i:i + chunk])), i + chunk >= 3000, True):
                if b >= 0:
                    opened.append(b)
                if e >= 0:
                    closed.append(e)
        assert [[b, e] for b, e in zip(opened, closed)] == whole


def test_detector_options_and_reset():
    o = VadOptions.from_conf({"max_end_silence_time": 500, "unknown_key": 1, "sil_pdf_ids": [0]})
    assert o.max_end_silence_time == 500 and o.speech_noise_thres == 0.6
    d = VadDetector(o)
    # 350 ms of silence now end a segment: frame 304 + 35 = 339, back by 35 - 10 - 1 = 24 -> frame 315 again
    assert d.step(sil_trace(500, [(100, 300)]), level(500), True, False) == [[740, 3160]]
    # a final call resets everything: the same input gives the same answer again
    assert d.step(sil_trace(500, [(100, 300)]), level(500), True, False) == [[740, 3160]]
    with pytest.raises(ValueError):
        d.step([0.5], [], False, True)


def test_a_level_below_the_threshold_is_silence():
    o = VadOptions(decibel_thres=-30.0)
    d = VadDetector(o)
    db = level(500)
    db[:] = -40.0
    assert d.step(sil_trace(500, [(100, 300)]), db, True, False) == []


# ---- the oracle's streaming pieces ---------------------------------------------------------------------------------
def test_oracle_cached_network_equals_one_shot():
    sd = vo.make_state_dict(seed=1)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((137, 400))
    whole, cache_whole = vo.fsmn_forward(sd, x)
    assert whole.shape == (137, 248) and np.allclose(whole.sum(axis=1), 1.0)
    cache = vo.new_cache()
    parts = []
    for a, b in ((0, 5), (5, 6), (6, 60), (60, 137)):
        s, cache = vo.fsmn_forward(sd, x[a:b], cache)
        parts.append(s)
    assert np.allclose(np.concatenate(parts), whole, atol=1e-12)
    assert np.allclose(cache, cache_whole, atol=1e-12)
    # the memory blocks look back 19 rows each: a change of row 10 reaches rows 10 .. 10 + 4 * 19 and no other
    x2 = x.copy(); x2[10] += 1.0
    lg, _ = vo.fsmn_forward(sd, x, return_logits=True)
    lg2, _ = vo.fsmn_forward(sd, x2, return_logits=True)
    changed = np.where(np.abs(lg - lg2).max(axis=1) > 1e-9)[0]
    assert changed.min() == 10 and changed.max() == 10 + 4 * 19    # four blocks of 19 rows of memory


def test_oracle_online_frontend_equals_offline():
    pcm = np.asarray(fo.synth_pcm("bursts", 16000 * 3 + 123, 7), dtype=np.int16)
    mel = fo.kaldi_fbank(pcm)
    want = vo.lfr_offline(mel)
    want_db = vo.frame_decibel(pcm)
    for chunk in (6400, 1600, 333, 100):
        fe = vo.OnlineFrontend()
        rows, dbs = [], []
        pieces = [pcm[i:i + chunk] for i in range(0, len(pcm), chunk)]
        for i, piece in enumerate(pieces):
            f, d = fe.push(piece, i == len(pieces) - 1)
            rows.append(f); dbs.append(d)
            assert f.shape[0] == d.shape[0]
        got = np.concatenate(rows)
        assert got.shape == want.shape, chunk
        assert np.allclose(got, want, atol=1e-9), chunk
        assert np.allclose(np.concatenate(dbs), want_db[: got.shape[0]], atol=1e-9)
    # non-final pushes hold back the rows whose right context is missing
    fe = vo.OnlineFrontend()
    f, _ = fe.push(pcm[:6400], False)
    assert f.shape[0] == fo.num_frames(6400) - 2


def test_energy_network_separates_loud_from_quiet():
    pcm = vu.bursts(16000 * 5, [(16000, 48000)], seed=1)
    sil, db = vu.oracle_sil_trace(pcm, vu.energy_state_dict())
    assert sil.shape == db.shape == (498,)
    assert sil[:95].min() > 0.8 and sil[105:295].max() < 0.05 and sil[305:].min() > 0.8
    assert db[150] > db[50] + 40
    segs = VadDetector().step(sil, db, True, False)
    assert len(segs) == 1 and abs(segs[0][0] - 740) <= 30 and abs(segs[0][1] - 3160) <= 30
