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
