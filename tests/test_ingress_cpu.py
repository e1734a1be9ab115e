"""Host side of the PCM ingress (WAV/base64 decoding, the int16 hand-off) against the numpy restatement of the
reference's functions.  Parity unpinned: the reference holds no fixtures for them."""
import base64
import importlib
import io
import wave

import numpy as np
import pytest

from oracle import ingress_oracle as io_oracle

ing = importlib.import_module("toolbox_for_asr_and_tts_amd.ingress")


def make_wav_b64(samples: np.ndarray, rate: int, channels: int) -> str:
    buf = io.BytesIO()
    with wave.open(buf, "wb") as wf:
        wf.setnchannels(channels)
        wf.setsampwidth(samples.dtype.itemsize)
        wf.setframerate(rate)
        wf.writeframes(samples.tobytes())
    return base64.b64encode(buf.getvalue()).decode("ascii")


@pytest.mark.parametrize("dtype,rate,channels", [
    (np.int16, 16000, 1), (np.int16, 16000, 2), (np.int16, 8000, 1), (np.int16, 48000, 2), (np.int16, 44100, 1),
    (np.int32, 16000, 1), (np.int32, 22050, 2), (np.uint8, 16000, 1), (np.uint8, 8000, 2)])
def test_decode_matches_the_reference_restatement(dtype, rate, channels):
    rng = np.random.default_rng(rate + channels)
    info = np.iinfo(dtype)
    x = rng.integers(info.min, info.max, size=(rate // 5) * channels, endpoint=True).astype(dtype)
    b64 = make_wav_b64(x, rate, channels)
    got, sr = ing.decode_wav_base64("  " + b64 + "\n")
    want, sr_ref = io_oracle.decode_wav_base64(b64)
    assert sr == sr_ref == 16000
    assert got.dtype == np.float32 and got.shape == want.shape
    assert abs(got.size - 3200) <= 1
    assert np.array_equal(got, want)


def test_decode_known_answers():
    x = np.array([0, 16384, -32768, 32767, 1, -1], np.int16)
    got, _ = ing.decode_wav_base64(make_wav_b64(x, 16000, 1))
    assert np.array_equal(got, np.array([0, 0.5, -1.0, 32767 / 32768, 1 / 32768, -1 / 32768], np.float32))
    st = np.array([100, 300, -32768, 32767], np.int16)     # two stereo frames -> means 200 and -0.5
    got, _ = ing.decode_wav_base64(make_wav_b64(st, 16000, 2))
    assert np.array_equal(got, np.array([200 / 32768, -0.5 / 32768], np.float32))
    # 8-bit: the reference subtracts in uint8, so codes below 128 wrap upwards
    u = np.array([128, 255, 127, 0], np.uint8)
    got, _ = ing.decode_wav_base64(make_wav_b64(u, 16000, 1))
    assert np.array_equal(got, np.array([0.0, 127 / 128, 255 / 128, 1.0], np.float32))


def test_decode_refuses_garbage():
    with pytest.raises(RuntimeError):
        ing.decode_wav_base64("")
    with pytest.raises(RuntimeError):
        ing.decode_wav_base64(base64.b64encode(b"not a wav file at all").decode())


def test_wire_int16_host_matches_the_restatement_and_truncates():
    edge = np.array([0.0, 1.0, -1.0, 1.5, -7.0, 0.5, -0.5, 3.0517578125e-05, -3.0517578125e-05, 0.99999, -0.99999,
                     1e-9, -1e-9, np.nextafter(np.float32(1), np.float32(0))], np.float32)
    want = np.array([0, 32767, -32767, 32767, -32767, 16383, -16383, 0, 0, 32766, -32766, 0, 0, 32766], np.int16)
    assert np.array_equal(ing.wire_int16(edge), want)
    assert np.array_equal(io_oracle.wire_int16(edge), want)
    x = np.random.default_rng(0).normal(0, 0.6, 100000).astype(np.float32)
    assert np.array_equal(ing.wire_int16(x), io_oracle.wire_int16(x))
    import torch
    assert np.array_equal(ing.wire_int16(torch.from_numpy(x)).numpy(), io_oracle.wire_int16(x))
    with pytest.raises(ValueError):
        ing.wire_int16(torch.zeros(4, dtype=torch.float64))


def test_accumulator_on_the_host_device_equals_concatenation():
    rng = np.random.default_rng(5)
    chunks = [rng.normal(0, 0.4, n).astype(np.float32) for n in (6400, 1, 0, 6400, 12345, 3)]
    acc = ing.PcmAccumulator(device="cpu", capacity=100)      # forces several growth steps
    for c in chunks:
        acc.append(c)
    want = io_oracle.accumulate(chunks)
    assert len(acc) == want.size
    assert np.array_equal(acc.samples().numpy(), want)
    pcm, lens = acc.wire()
    assert pcm.shape == (1, want.size) and int(lens[0]) == want.size
    assert np.array_equal(pcm[0].numpy(), io_oracle.wire_int16(want))
    acc.clear()
    assert len(acc) == 0
    acc.append(chunks[0])
    assert np.array_equal(acc.samples().numpy(), chunks[0])


def _scripted_chunks(seed, n_chunks):
    """Chunks of 240-400 ms around the gate's two thresholds: loud, quiet, loud-mean-but-low-peak, spiky-but-quiet."""
    rng = np.random.default_rng(seed)
    out = []
    t = 100.0
    for _ in range(n_chunks):
        n = int(rng.choice([3840, 6400, 6400, 1000]))
        kind = rng.choice(["loud", "quiet", "flat", "spike", "edge"], p=[0.35, 0.35, 0.1, 0.1, 0.1])
        if kind == "loud":
            x = rng.normal(0, 0.2, n)
        elif kind == "quiet":
            x = rng.normal(0, 0.004, n)
        elif kind == "flat":                       # mean |x| above 0.03, peak below 0.17
            x = rng.uniform(-0.12, 0.12, n)
        elif kind == "spike":                      # peak above 0.17, mean below 0.03
            x = rng.normal(0, 0.002, n)
            x[rng.integers(0, n)] = 0.9
        else:                                      # mean right at the threshold
            x = np.full(n, 0.03) * rng.choice([-1, 1], n)
            x[0] = 0.5
            x[1:] *= rng.choice([0.9999, 1.0001])
        t += n / 16000 + float(rng.choice([0.0, 0.0, 0.7, 1.4]))
        out.append((x.astype(np.float32), t))
    return out


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_streaming_session_buffer_follows_the_reference_policy(seed):
    ref = io_oracle.SessionBuffer(now=100.0)
    ses = ing.StreamingPcmSession(device="cpu", capacity=64, now=100.0)
    finals = 0
    for x, t in _scripted_chunks(seed, 60):
        want = ref.process_chunk(x, t)
        got = ses.process_chunk(x, t)
        assert got == want
        assert len(ses.audio) == len(ref.audio_buffer)
        if got["should_finalize"]:
            finals += 1
            assert np.array_equal(ses.samples().numpy(), ref.audio_buffer)
            assert np.array_equal(ses.wire()[0][0].numpy(), io_oracle.wire_int16(ref.audio_buffer))
            ref = io_oracle.SessionBuffer(now=t)
            ses.reset(now=t)
    assert np.array_equal(ses.samples().numpy(), ref.audio_buffer)
    assert finals >= 1


def test_streaming_session_known_sequence():
    """Silence, silence, speech, silence x3, then a long gap: the newest 400 ms before the speech are kept, only two
    trailing silent chunks are, and finalize fires once 2 s have passed since the last speech chunk."""
    q = lambda v: np.full(6400, v, np.float32)
    ses = ing.StreamingPcmSession(device="cpu", now=0.0)
    assert ses.process_chunk(q(0.001), 0.4) == {"is_speech": False, "intermediate_text": "", "should_finalize": False}
    ses.process_chunk(q(0.002), 0.8)
    assert len(ses.audio) == 0
    r = ses.process_chunk(q(0.3), 1.2)
    assert r["is_speech"] and len(ses.audio) == 12800
    assert float(ses.samples()[0]) == np.float32(0.002) and float(ses.samples()[6400]) == np.float32(0.3)
    for k, t in enumerate((1.6, 2.0, 2.4)):
        r = ses.process_chunk(q(0.003), t)
        assert not r["is_speech"] and not r["should_finalize"]
    assert len(ses.audio) == 12800 + 2 * 6400
    assert ses.process_chunk(q(0.003), 3.19)["should_finalize"] is False
    assert ses.process_chunk(q(0.003), 3.2)["should_finalize"] is True
    assert len(ses.audio) == 12800 + 2 * 6400


def test_session_asks_the_vad_model_per_chunk_and_decides_by_energy():
    """voice_interface.py:1585-1606: generate(input=chunk, cache=session cache, is_final=False, chunk_size=ms); the answer is
    read from [{"value": [...]}] and the decision stays the energy rule's; a failing model changes nothing."""
    calls = []

    class FakeVad:
        def __init__(self):
            self.fail = False

        def generate(self, input, cache, is_final, chunk_size):
            if self.fail:
                raise RuntimeError("model down")
            calls.append((len(input), id(cache), is_final, chunk_size))
            return [{"key": "k", "value": [[0, -1]] if np.abs(input).max() > 0.01 else []}]

    vad = FakeVad()
    ses = ing.StreamingPcmSession(device="cpu", now=0.0, vad_model=vad)
    quiet = np.full(6400, 0.02, np.float32)          # the model says speech, the energy rule says no
    st = ses.process_chunk(quiet, now=0.4)
    assert st["is_speech"] is False and ses.is_speech_vad is True
    loud = np.full(6400, 0.2, np.float32)
    st = ses.process_chunk(loud, now=0.8)
    assert st["is_speech"] is True and ses.is_speech_vad is True
    assert ses.process_chunk(np.zeros(3200, np.float32), now=1.0)["is_speech"] is False and ses.is_speech_vad is False
    assert [c[0] for c in calls] == [6400, 6400, 3200] and [c[3] for c in calls] == [400, 400, 200]
    assert all(c[2] is False for c in calls) and len({c[1] for c in calls}) == 1      # one cache for the session
    cache_before = calls[-1][1]
    ses.reset(now=2.0)
    ses.process_chunk(loud, now=2.4)
    assert calls[-1][1] != cache_before               # reset starts a new VAD stream
    vad.fail = True
    st = ses.process_chunk(loud, now=2.8)
    assert st["is_speech"] is True and ses.is_speech_vad is False
