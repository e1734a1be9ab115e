"""CPU restatement of the reference's PCM ingress around the pipeline call.  TEST INFRASTRUCTURE ONLY: nothing
in the product path imports this module.

Parity unpinned: the reference holds no fixtures for these two functions; they are restated line by line from

  * voice-service/app/services/voice_interface.py:982-1051  ``base64_to_audio_np`` (wave-module branch): WAV bytes ->
    float in [-1, 1) by sample width, channel mean, FFT resampling to 16 kHz, float32;
  * voice-service/app/services/voice_interface.py:1997-2010  ``StreamingASRSession.finalize``: the float32 session
    buffer is clipped to [-1, 1], multiplied by 32767 and converted to int16 (numpy's conversion truncates toward
    zero); that int16 WAV is what the recognition pipeline reads (:2040-2053), so it is the wire type of the hot path.
"""
from __future__ import annotations

import base64
import wave
from io import BytesIO

import numpy as np

TARGET_SR = 16000  # voice_interface.py:647


def decode_wav_base64(b64: str):
    """voice_interface.py:988-1051 (the branch that needs no torchaudio)."""
    audio_bytes = base64.b64decode(b64.strip())
    if not audio_bytes:
        raise ValueError("empty audio")
    with wave.open(BytesIO(audio_bytes), "rb") as wf:
        sr, ch, sw, n = wf.getframerate(), wf.getnchannels(), wf.getsampwidth(), wf.getnframes()
        raw = wf.readframes(n)
    if sw == 1:
        a = np.frombuffer(raw, dtype=np.uint8)
        a = (a - 128) / 128.0  # uint8 arithmetic wraps below 128, exactly as in the reference (:1005-1006)
    elif sw == 2:
        a = np.frombuffer(raw, dtype=np.int16) / 32768.0
    elif sw == 4:
        a = np.frombuffer(raw, dtype=np.int32) / 2147483648.0
    else:
        raise ValueError(f"unsupported sample width {sw}")
    if ch > 1:
        a = np.mean(a.reshape(-1, ch), axis=1)
    if sr != TARGET_SR:
        from scipy import signal
        a = signal.resample(a, int(len(a) * TARGET_SR / sr))
    return a.astype(np.float32), TARGET_SR


def wire_int16(buf: np.ndarray) -> np.ndarray:
    """voice_interface.py:1997-2010: clip, x 32767 in float32, truncate to int16."""
    buf = np.asarray(buf, dtype=np.float32)
    return (np.clip(buf, -1.0, 1.0) * 32767.0).astype(np.int16)


def accumulate(chunks) -> np.ndarray:
    """voice_interface.py:1707: the session buffer is the concatenation of the float32 chunks."""
    out = np.array([], dtype=np.float32)
    for c in chunks:
        out = np.concatenate([out, np.asarray(c, dtype=np.float32)])
    return out
