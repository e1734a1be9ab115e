"""PCM ingress of the hot path: what the reference does to audio between the socket and the recognition pipeline.

* ``decode_wav_base64`` stands in for ``base64_to_audio_np``
  (/root/reference/voice-service/app/services/voice_interface.py:982-1051, wave-module branch): WAV bytes, base64
  coded, to mono float32 in [-1, 1) at 16 kHz.  Host code, as in the reference (a few kB per 400 ms chunk).
* ``wire_int16`` is the hand-off of ``StreamingASRSession.finalize`` (voice_interface.py:1997-2010): clip to
  [-1, 1], times 32767, truncate to int16.  The reference writes that int16 signal to a WAV file which the pipeline
  reads back (:2011-2053); here it stays a device tensor and goes straight into ``asrfe_frontend`` (its int16 input
  is exactly that wire type).
* ``PcmAccumulator`` replaces the session's ``np.concatenate([buffer, chunk])`` per chunk (:1688-1742, quadratic
  in the utterance length) by a device-resident buffer that grows geometrically.
"""
from __future__ import annotations

import base64
import wave
from io import BytesIO

import numpy as np
import torch

TARGET_SAMPLE_RATE = 16000  # STREAMING_TARGET_SAMPLE_RATE, voice_interface.py:647


def decode_wav_base64(base64_str: str):
    """Returns (float32 mono samples, 16000).  Sample widths 1, 2 and 4 bytes; channels are averaged; other rates are
    resampled with ``scipy.signal.resample`` as the reference does.  8-bit input keeps the reference's unsigned
    wrap-around below 128 (voice_interface.py:1005-1006 subtracts in uint8): results are identical, quirk included."""
    audio_bytes = base64.b64decode(base64_str.strip())
    if not audio_bytes:
        raise RuntimeError("audio decoding failed: empty payload")
    try:
        with wave.open(BytesIO(audio_bytes), "rb") as wf:
            rate, channels, width = wf.getframerate(), wf.getnchannels(), wf.getsampwidth()
            raw = wf.readframes(wf.getnframes())
    except (wave.Error, EOFError) as e:
        raise RuntimeError(f"audio decoding failed: not a PCM WAV ({e})") from e
    if width == 1:
        x = np.frombuffer(raw, dtype=np.uint8)
        x = (x - np.uint8(128)) / 128.0
    elif width == 2:
        x = np.frombuffer(raw, dtype=np.int16) / 32768.0
    elif width == 4:
        x = np.frombuffer(raw, dtype=np.int32) / 2147483648.0
    else:
        raise RuntimeError(f"audio decoding failed: unsupported sample width {width}")
    if channels > 1:
        x = x.reshape(-1, channels).mean(axis=1)
    if rate != TARGET_SAMPLE_RATE:
        from scipy import signal
        x = signal.resample(x, int(len(x) * TARGET_SAMPLE_RATE / rate))
    return x.astype(np.float32), TARGET_SAMPLE_RATE


def wire_int16(samples):
    """float32 samples (numpy array or tensor on any device) -> the int16 signal the reference hands to the pipeline:
    ``(clip(x, -1, 1) * 32767).astype(int16)``, computed in float32, truncating toward zero."""
    if isinstance(samples, torch.Tensor):
        if samples.dtype != torch.float32:
            raise ValueError("wire_int16: float32 samples expected")
        return samples.clamp(-1.0, 1.0).mul(32767.0).to(torch.int16)
    x = np.asarray(samples, dtype=np.float32)
    return (np.clip(x, -1.0, 1.0) * np.float32(32767.0)).astype(np.int16)


class PcmAccumulator:
    """Device-resident float32 session buffer: ``append`` copies a chunk behind the samples already held (capacity
    doubles when it runs out, so n chunks cost O(total samples), not O(n x total)); ``wire`` returns the int16
    hand-off of everything held, shaped ``[1, n]`` with its length tensor, ready for ``AsrEngine`` / ``WavFrontend``."""

    def __init__(self, device="cuda", capacity: int = 16000 * 30):
        self.device = torch.device(device)
        self._buf = torch.empty((max(1, int(capacity)),), dtype=torch.float32, device=self.device)
        self._n = 0

    def __len__(self) -> int:
        return self._n

    def clear(self):
        self._n = 0

    def append(self, chunk):
        c = torch.as_tensor(chunk, dtype=torch.float32).reshape(-1)
        need = self._n + c.numel()
        if need > self._buf.numel():
            grown = torch.empty((max(need, 2 * self._buf.numel()),), dtype=torch.float32, device=self.device)
            grown[: self._n].copy_(self._buf[: self._n])
            self._buf = grown
        self._buf[self._n:need].copy_(c, non_blocking=True)
        self._n = need

    def samples(self) -> torch.Tensor:
        """The float32 samples held so far (a view)."""
        return self._buf[: self._n]

    def wire(self):
        """(int16 [1, n] on the device, int32 lengths [1] on the device)."""
        pcm = wire_int16(self._buf[: self._n]).reshape(1, -1)
        return pcm, torch.tensor([self._n], dtype=torch.int32, device=self.device)


class StreamingPcmSession:
    """The part of the reference's ``StreamingASRSession`` that decides which samples reach the recognizer
    (voice_interface.py: state :1103-1116, ``process_chunk`` :1569-1578, :1606, :1672-1747, :1816-1818), with the
    utterance held on the device:

    * a chunk counts as speech when mean|x| > 0.03 and max|x| > 0.17.  The reference also asks its VAD model per chunk
      (``vad_model.generate(input=chunk, cache=self.vad_cache, is_final=False, chunk_size=ms)``, :1585-1600) and then
      decides by the energy rule alone (:1605-1606).  With ``vad_model`` given (``AutoModel`` of an FSMN-VAD directory)
      the same call is made here, a failing call falls back to the energy rule as there (:1601-1602), and the answer is
      kept in ``is_speech_vad``; the decision stays the energy rule's;
    * before the first speech chunk only the newest 400 ms are kept (host side, at most 6400 samples) and they are
      put in front of that first speech chunk;
    * after speech, every speech chunk and at most two silent chunks in a row are appended;
    * ``should_finalize`` once 2 s have passed since the last speech chunk.

    ``process_chunk`` returns the reference's dictionary (``intermediate_text`` is always empty there too).  ``now``
    defaults to ``time.time()`` as in the reference; tests pass their own clock.  KWS, speaker verification and the
    recognizer call itself are not part of this class: ``wire()`` / ``samples()`` give what ``finalize`` would send."""
    VAD_ENERGY_THRESHOLD = 0.03
    VAD_MAX_THRESHOLD = 0.17
    SILENCE_THRESHOLD = 2.0
    PRE_SPEECH_MAX_DURATION = 0.4

    def __init__(self, device="cuda", capacity: int = 16000 * 30, now: float | None = None, vad_model=None):
        import time
        self._clock = time.time
        self.audio = PcmAccumulator(device, capacity)
        self.vad_model = vad_model
        self.reset(now)

    def reset(self, now: float | None = None):
        self.audio.clear()
        self._pre = np.zeros((0,), np.float32)
        self.silence_timer = 0.0
        self.last_voice_time = self._clock() if now is None else now
        self.has_detected_speech = False
        self.silence_chunk_count = 0
        self.vad_cache = {}           # the reference clears it with the session state (voice_interface.py:1157, 1262)
        self.is_speech_vad = False

    def _ask_vad(self, x) -> bool:
        try:
            res = self.vad_model.generate(input=x, cache=self.vad_cache, is_final=False,
                                          chunk_size=int(len(x) / TARGET_SAMPLE_RATE * 1000))
            if isinstance(res, list) and res and isinstance(res[0], dict):
                value = res[0].get("value", [])
                if isinstance(value, list):
                    return len(value) > 0
                if isinstance(value, str):
                    return value.lower() == "speech"
        except Exception:  # the reference logs a warning and goes on with the energy rule
            pass
        return False

    def process_chunk(self, audio_np, now: float | None = None):
        now = self._clock() if now is None else now
        x = np.asarray(audio_np)
        mag = np.abs(x)
        is_speech = bool(np.mean(mag) > self.VAD_ENERGY_THRESHOLD and np.max(mag) > self.VAD_MAX_THRESHOLD)
        self.is_speech_vad = self._ask_vad(x) if self.vad_model is not None and x.size else False
        if is_speech:
            self.silence_timer = 0.0
            self.last_voice_time = now
            self.has_detected_speech = True
            self.silence_chunk_count = 0
            if self._pre.size:
                self.audio.append(self._pre)
                self._pre = np.zeros((0,), np.float32)
            self.audio.append(x)
        elif self.has_detected_speech:
            if self.silence_chunk_count < 2:
                self.audio.append(x)
                self.silence_chunk_count += 1
            self.silence_timer = now - self.last_voice_time
        else:
            keep = int(self.PRE_SPEECH_MAX_DURATION * TARGET_SAMPLE_RATE)
            self._pre = np.concatenate([self._pre, x.astype(np.float32, copy=False)])[-keep:]
            self.silence_timer = 0.0
        should_finalize = bool(self.silence_timer >= self.SILENCE_THRESHOLD and len(self.audio) > 0
                               and self.has_detected_speech)
        return {"is_speech": is_speech, "intermediate_text": "", "should_finalize": should_finalize}

    def samples(self) -> torch.Tensor:
        return self.audio.samples()

    def wire(self):
        return self.audio.wire()
