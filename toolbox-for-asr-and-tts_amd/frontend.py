"""Host-side mirror of FunASR's ``WavFrontend`` over the libasrfe HIP kernels.

The reference reaches ``WavFrontend.forward(input, input_lengths)`` inside
``funasr.AutoModel`` / the ModelScope ASR pipeline
(/root/reference/voice-service/app/services/voice_interface.py:2049-2053, :1585-1590,
:1370-1374).  Names, argument meaning and return values follow that class so that a caller
can swap it in: ``forward`` returns ``(feats_pad [B, T', n_mels * lfr_m], feats_lens [B])``.
All arithmetic runs in the HIP library; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

_WINDOWS = {"hamming": 0, "hanning": 1, "hann": 1, "povey": 2, "rectangular": 3}


def _max0(t: torch.Tensor) -> int:
    return int(t.max()) if t.numel() else 0


def load_cmvn(cmvn_file: str):
    """Parses a Kaldi-nnet ``am.mvn``: returns (shift, scale) float32 arrays (<AddShift>, <Rescale>)."""
    with open(cmvn_file, "r", encoding="utf-8") as f:
        lines = f.readlines()
    shift = scale = None
    for i, line in enumerate(lines):
        items = line.split()
        if not items:
            continue
        if items[0] in ("<AddShift>", "<Rescale>") and i + 1 < len(lines):
            nxt = lines[i + 1].split()
            if nxt and nxt[0] == "<LearnRateCoef>":
                vals = np.array([float(v) for v in nxt[3:len(nxt) - 1]], dtype=np.float32)
                if items[0] == "<AddShift>":
                    shift = vals
                else:
                    scale = vals
    if shift is None or scale is None:
        raise ValueError(f"{cmvn_file}: no <AddShift>/<Rescale> rows found")
    return shift, scale


class WavFrontend:
    """fbank (Kaldi) -> LFR -> CMVN on the GPU.

    Arguments as in FunASR: ``frame_length`` / ``frame_shift`` in milliseconds.  ``dither`` must be 0
    (the kernels are deterministic); ``upsacle_samples`` (sic) keeps FunASR's spelling and means float
    PCM in [-1, 1) is multiplied by 32768, which is what the int16 input path is equal to.
    ``session_float`` (no FunASR counterpart) treats float32 input as the reference's streaming-session buffer
    instead: the kernel quantises it on load exactly as the session's hand-off does, int16(clip(x, -1, 1) * 32767)
    truncating (voice_interface.py:1997-2010), so the features equal those of the int16 WAV the reference writes.
    """

    def __init__(self, cmvn_file: str | None = None, fs: int = 16000, window: str = "hamming", n_mels: int = 80,
                 frame_length: int = 25, frame_shift: int = 10, lfr_m: int = 7, lfr_n: int = 6, dither: float = 0.0,
                 snip_edges: bool = True, upsacle_samples: bool = True, cmvn: tuple | None = None,
                 ctx: _lib.Context | None = None, session_float: bool = False, **kwargs):
        self.session_float = bool(session_float)
        if dither != 0.0:
            raise ValueError("WavFrontend: only dither=0.0 is supported")
        if not snip_edges:
            raise ValueError("WavFrontend: only snip_edges=True is supported")
        if not upsacle_samples:
            raise ValueError("WavFrontend: only upsacle_samples=True is supported")
        if window not in _WINDOWS:
            raise ValueError(f"WavFrontend: unknown window {window!r}")
        self.fs, self.window, self.n_mels = fs, window, n_mels
        self.frame_length, self.frame_shift = frame_length, frame_shift
        self.lfr_m, self.lfr_n = lfr_m, lfr_n
        if ctx is None:
            cfg = _lib.default_config()
            cfg.sample_rate = fs
            cfg.frame_length = fs * frame_length // 1000
            cfg.frame_shift = fs * frame_shift // 1000
            cfg.n_mels = n_mels
            cfg.window_type = _WINDOWS[window]
            cfg.lfr_m, cfg.lfr_n = lfr_m, lfr_n
            ctx = _lib.Context(cfg)
        self.ctx = ctx
        self.cmvn_file = cmvn_file
        if cmvn_file is not None:
            cmvn = load_cmvn(cmvn_file)
        if cmvn is not None:
            self.set_cmvn(*cmvn)

    # ---- configuration -------------------------------------------------------------------
    def output_size(self) -> int:
        return self.n_mels * self.lfr_m

    def set_cmvn(self, shift, scale):
        shift = np.ascontiguousarray(np.asarray(shift, dtype=np.float32))
        scale = np.ascontiguousarray(np.asarray(scale, dtype=np.float32))
        if shift.shape != scale.shape or shift.ndim != 1:
            raise ValueError("cmvn shift/scale must be 1-D arrays of equal length")
        fp = C.POINTER(C.c_float)
        self.ctx.check(self.ctx.lib.asrfe_load_cmvn(self.ctx.handle, shift.ctypes.data_as(fp),
                                                    scale.ctypes.data_as(fp), shift.shape[0]))

    # ---- helpers ---------------------------------------------------------------------------
    def num_frames(self, n_samples: int) -> int:
        return int(self.ctx.lib.asrfe_num_frames(self.ctx.handle, int(n_samples)))

    def num_lfr_frames(self, n_frames: int) -> int:
        return int(self.ctx.lib.asrfe_num_lfr_frames(self.ctx.handle, int(n_frames)))

    def _prep(self, input: torch.Tensor, input_lengths):
        if not isinstance(input, torch.Tensor) or not input.is_cuda:
            raise ValueError("WavFrontend: input must be a CUDA/HIP tensor [B, N]")
        if input.dim() == 1:
            input = input[None]
        if input.dim() != 2:
            raise ValueError("WavFrontend: input must be [B, N]")
        if input.dtype == torch.int16:
            dt = _lib.PCM_I16
        elif input.dtype == torch.float32:
            dt = _lib.PCM_F32_WIRE if self.session_float else _lib.PCM_F32
        else:
            raise ValueError("WavFrontend: PCM dtype must be int16 or float32")
        if input.stride(1) != 1 and input.shape[1] > 1:
            input = input.contiguous()
        lens = torch.as_tensor(input_lengths)
        lens_host = lens.detach().to("cpu", torch.int64)
        if lens_host.numel() != input.shape[0]:
            raise ValueError("WavFrontend: input_lengths must have one entry per row")
        if _max0(lens_host) > input.shape[1] or (lens_host.numel() and int(lens_host.min()) < 0):
            raise ValueError("WavFrontend: input_lengths out of range")
        lens_dev = lens_host.to(torch.int32).to(input.device)
        return input, dt, lens_host, lens_dev

    @staticmethod
    def _stream():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # ---- the three FunASR entry points ---------------------------------------------------------
    def forward_fbank(self, input: torch.Tensor, input_lengths):
        """kaldi.fbank only: returns (logmel [B, Tmax, n_mels] float32, n_frames [B] int32)."""
        input, dt, lens_host, lens_dev = self._prep(input, input_lengths)
        b = input.shape[0]
        t_max = max(1, self.num_frames(_max0(lens_host)))
        out = torch.empty((b, t_max, self.n_mels), dtype=torch.float32, device=input.device)
        n_frames = torch.zeros((b,), dtype=torch.int32, device=input.device)
        if b and input.shape[1] > 0:
            with torch.cuda.device(input.device):
                self.ctx.check(self.ctx.lib.asrfe_fbank(self.ctx.handle, input.data_ptr(), dt, input.stride(0),
                                                        lens_dev.data_ptr(), b, t_max, out.data_ptr(),
                                                        n_frames.data_ptr(), self._stream()))
        t_true = self.num_frames(_max0(lens_host))
        return out[:, :t_true], n_frames

    def forward_lfr_cmvn(self, input: torch.Tensor, input_lengths, out_dtype=torch.float32):
        """apply_lfr + apply_cmvn on log-mel [B, T, n_mels]: returns (feats [B, T', n_mels*lfr_m], lens [B])."""
        if not input.is_cuda or input.dtype != torch.float32 or input.dim() != 3:
            raise ValueError("forward_lfr_cmvn: input must be a float32 CUDA tensor [B, T, n_mels]")
        input = input.contiguous()
        b, t_max, _ = input.shape
        lens_host = torch.as_tensor(input_lengths).detach().to("cpu", torch.int64)
        lens_dev = lens_host.to(torch.int32).to(input.device)
        tp_true = self.num_lfr_frames(_max0(lens_host))
        tp_max = max(1, tp_true)
        feats = torch.empty((b, tp_max, self.output_size()), dtype=out_dtype, device=input.device)
        lens_out = torch.empty((b,), dtype=torch.int32, device=input.device)
        if b and t_max:
            with torch.cuda.device(input.device):
                self.ctx.check(self.ctx.lib.asrfe_lfr_cmvn(
                    self.ctx.handle, input.data_ptr(), lens_dev.data_ptr(), b, t_max, tp_max, feats.data_ptr(),
                    _lib.DT_BF16 if out_dtype == torch.bfloat16 else _lib.DT_F32, lens_out.data_ptr(), self._stream()))
        else:
            lens_out.zero_()
        return feats[:, :tp_true], lens_out

    def forward(self, input: torch.Tensor, input_lengths, out_dtype=torch.float32, **kwargs):
        """PCM [B, N] (int16 or float32) -> (feats_pad [B, T', n_mels*lfr_m], feats_lens [B] int32), one fused kernel."""
        input, dt, lens_host, lens_dev = self._prep(input, input_lengths)
        b = input.shape[0]
        tp_true = self.num_lfr_frames(self.num_frames(_max0(lens_host)))
        tp_max = max(1, tp_true)
        feats = torch.empty((b, tp_max, self.output_size()), dtype=out_dtype, device=input.device)
        lens_out = torch.zeros((b,), dtype=torch.int32, device=input.device)
        if b and input.shape[1] > 0:
            with torch.cuda.device(input.device):
                self.ctx.check(self.ctx.lib.asrfe_frontend(
                    self.ctx.handle, input.data_ptr(), dt, input.stride(0), lens_dev.data_ptr(), b, tp_max,
                    feats.data_ptr(), _lib.DT_BF16 if out_dtype == torch.bfloat16 else _lib.DT_F32,
                    lens_out.data_ptr(), self._stream()))
        return feats[:, :tp_true], lens_out

    __call__ = forward
