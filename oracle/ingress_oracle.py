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


class SessionBuffer:
    """voice_interface.py ``StreamingASRSession``: only the state that decides WHICH samples reach the recognizer
    (init :1103-1116, energy gate :1569-1578 and :1606, accumulation policy :1672-1747, finalize condition
    :1816-1818).  The clock is an argument so the restatement is deterministic.  The VAD model's answer is computed
    by the reference but not used (:1605-1606), so it is absent here."""
    ENERGY_THRESHOLD = 0.03   # :656
    MAX_THRESHOLD = 0.17      # :657
    SILENCE_THRESHOLD = 2.0   # :649
    PRE_SPEECH_SECONDS = 0.4  # :1116

    def __init__(self, now: float):
        self.audio_buffer = np.array([], dtype=np.float32)
        self.pre_speech_buffer = np.array([], dtype=np.float32)
        self.silence_timer = 0.0
        self.last_voice_time = now
        self.has_detected_speech = False
        self.silence_chunk_count = 0

    def process_chunk(self, audio_np: np.ndarray, now: float):
        audio_energy = np.mean(np.abs(audio_np))
        audio_max = np.max(np.abs(audio_np))
        is_speech = bool(audio_energy > self.ENERGY_THRESHOLD and audio_max > self.MAX_THRESHOLD)
        if is_speech:
            self.silence_timer = 0.0
            self.last_voice_time = now
            self.has_detected_speech = True
            self.silence_chunk_count = 0
            if len(self.pre_speech_buffer) > 0:
                self.audio_buffer = np.concatenate([self.audio_buffer, self.pre_speech_buffer])
                self.pre_speech_buffer = np.array([], dtype=np.float32)
            self.audio_buffer = np.concatenate([self.audio_buffer, audio_np])
        elif self.has_detected_speech:
            if self.silence_chunk_count < 2:
                self.audio_buffer = np.concatenate([self.audio_buffer, audio_np])
                self.silence_chunk_count += 1
            self.silence_timer = now - self.last_voice_time
        else:
            self.pre_speech_buffer = np.concatenate([self.pre_speech_buffer, audio_np])
            target = int(self.PRE_SPEECH_SECONDS * TARGET_SR)
            if len(self.pre_speech_buffer) > target:
                self.pre_speech_buffer = self.pre_speech_buffer[-target:]
            self.silence_timer = 0.0
        should_finalize = bool(self.silence_timer >= self.SILENCE_THRESHOLD and len(self.audio_buffer) > 0
                               and self.has_detected_speech)
        return {"is_speech": is_speech, "intermediate_text": "", "should_finalize": should_finalize}
