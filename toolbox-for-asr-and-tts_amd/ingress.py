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
