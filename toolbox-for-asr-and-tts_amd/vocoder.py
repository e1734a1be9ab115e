"""Host-side mirror of the HiFi-GAN generator over libasrfe (mel spectrogram -> waveform).

The reference's TTS path ends in ``pipeline_instance(input=seg, voice=voice, forward_params=...)``
(/root/reference/tts-service/app/services/tts_service.py:617-625), ModelScope's Sambert-HiFiGAN pipeline
(tts_service.py:34-38, 164-177), whose last stage is this vocoder; its output contract is a 16-bit mono WAV image
(tts_service.py:396-411).  ``HiFiGANGenerator`` takes the published generator's ``state_dict`` (weight norm removed,
or ``weight_g`` / ``weight_v`` pairs, which are folded here) and runs every convolution on the device; the acoustic
model that produces the mel spectrogram from text is not part of this package.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import io
import wave

import numpy as np
import torch

from . import _lib


class HiFiGANGenerator:
    def __init__(self, ctx: _lib.Context | None = None, n_mels: int = 80, initial_channel: int = 512,
                 upsample_rates=(8, 8, 2, 2), upsample_kernels=(16, 16, 4, 4), resblock_kernels=(3, 7, 11),
                 resblock_dilations=((1, 3, 5), (1, 3, 5), (1, 3, 5)), lrelu_slope: float = 0.1, final_slope: float = 0.01,
                 sample_rate: int = 24000):
        self.ctx = ctx if ctx is not None else _lib.Context()
        cfg = _lib.VocoderConfig()
        rc = self.ctx.lib.asrfe_default_vocoder_config(C.byref(cfg))
        if rc != 0:
            raise _lib.AsrfeError(rc, "asrfe_default_vocoder_config")
        if len(upsample_rates) != len(upsample_kernels) or len(resblock_kernels) != len(resblock_dilations):
            raise ValueError("HiFiGANGenerator: rates / kernels and residual-block kernels / dilations must pair up")
        cfg.n_mels, cfg.initial_channel = n_mels, initial_channel
        cfg.n_upsamples, cfg.n_resblocks = len(upsample_rates), len(resblock_kernels)
        for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernels)):
            cfg.upsample_rates[i], cfg.upsample_kernels[i] = u, k
        for j, (k, dils) in enumerate(zip(resblock_kernels, resblock_dilations)):
            cfg.resblock_kernels[j] = k
            if len(dils) != 3:
                raise ValueError("HiFiGANGenerator: three dilations per residual block")
            for q, d in enumerate(dils):
                cfg.resblock_dilations[j][q] = d
        cfg.lrelu_slope, cfg.final_slope = lrelu_slope, final_slope
        self.cfg = cfg
        self.n_mels = n_mels
        self.sample_rate = sample_rate
        self.hop = int(np.prod(upsample_rates))
        self._ws = None

    def load_state_dict(self, state_dict):
        """Generator tensors under the published names; ``X.weight_g`` / ``X.weight_v`` pairs (weight norm still in
        place) are folded into ``X.weight = g * v / ||v||`` (norm over all but the first dimension)."""
        sd = {}
        for name, t in state_dict.items():
            t = t.detach().to("cpu", torch.float32) if isinstance(t, torch.Tensor) else torch.as_tensor(np.asarray(t), dtype=torch.float32)
            sd[name[len("generator."):] if name.startswith("generator.") else name] = t
        for name in [n for n in sd if n.endswith(".weight_v")]:
            base = name[: -len(".weight_v")]
            v, g = sd.pop(name), sd.pop(base + ".weight_g")
            norm = v.flatten(1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
            sd[base + ".weight"] = g.reshape_as(norm) * v / norm
        keep, descs = [], []
        for name, t in sd.items():
            a = t.contiguous().numpy()
            if a.ndim < 1 or a.ndim > 4:
                continue
            keep.append(a)
            d = _lib.TensorDesc()
            d.name = name.encode()
            d.data_host = a.ctypes.data_as(C.POINTER(C.c_float))
            d.ndim = a.ndim
            for i, s in enumerate(a.shape):
                d.shape[i] = s
            descs.append(d)
        arr = (_lib.TensorDesc * len(descs))(*descs)
        self.ctx.check(self.ctx.lib.asrfe_load_vocoder_weights(self.ctx.handle, C.byref(self.cfg), arr, len(descs)))
        assert int(self.ctx.lib.asrfe_vocoder_hop(self.ctx.handle)) == self.hop
        return self

    def workspace_bytes(self, batch: int, t_mel: int) -> int:
        n = int(self.ctx.lib.asrfe_vocoder_workspace_bytes(self.ctx.handle, batch, t_mel))
        if n < 0:
            raise ValueError("HiFiGANGenerator: weights not loaded or bad sizes")
        return n

    def forward(self, mel: torch.Tensor) -> torch.Tensor:
        """mel [B, n_mels, T] float32 on the device -> waveform [B, T * hop] float32 in (-1, 1)."""
        if not mel.is_cuda or mel.dtype != torch.float32 or mel.dim() != 3 or mel.shape[1] != self.n_mels:
            raise ValueError(f"HiFiGANGenerator: mel must be a float32 CUDA tensor [B, {self.n_mels}, T]")
        mel = mel.contiguous()
        b, _, t = mel.shape
        wav = torch.empty((b, t * self.hop), dtype=torch.float32, device=mel.device)
        if b and t:
            nbytes = self.workspace_bytes(b, t)
            if self._ws is None or self._ws.numel() < nbytes + 256 or self._ws.device != mel.device:
                self._ws = torch.empty((nbytes + 256,), dtype=torch.uint8, device=mel.device)
            ptr = (self._ws.data_ptr() + 255) // 256 * 256
            with torch.cuda.device(mel.device):
                self.ctx.check(self.ctx.lib.asrfe_vocoder_forward(
                    self.ctx.handle, mel.data_ptr(), b, t, wav.data_ptr(), ptr, self._ws.numel() - (ptr - self._ws.data_ptr()),
                    C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return wav

    __call__ = forward

    def wav_bytes(self, wav_row: torch.Tensor) -> bytes:
        """One waveform -> the image of a 16-bit mono WAV file at ``sample_rate`` (the reference's output contract)."""
        pcm = (wav_row.detach().float().clamp(-1.0, 1.0) * 32767.0).round().to(torch.int16).cpu().numpy()
        buf = io.BytesIO()
        with wave.open(buf, "wb") as wf:
            wf.setnchannels(1)
            wf.setsampwidth(2)
            wf.setframerate(self.sample_rate)
            wf.writeframes(pcm.tobytes())
        return buf.getvalue()
