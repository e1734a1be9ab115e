"""Host-side mirror of FunASR's ``SANMEncoder`` (Paraformer-large acoustic encoder) over libasrfe.

The reference reaches ``SANMEncoder.forward(xs_pad, ilens)`` inside the ModelScope ASR pipeline it
builds at /root/reference/voice-service/app/services/voice_interface.py:832-841 and calls at
:2049-2053.  Constructor arguments, ``state_dict`` key names (``encoders0.0.*``, ``encoders.N.*``,
``after_norm.*``) and the ``(xs_pad, olens, None)`` return follow that class.  All arithmetic runs in
the HIP library (fp16 MFMA operands by default, bf16 selectable, fp32 accumulation, fp32 residual stream);
there is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


# 16-bit type of the MFMA operands.  float16 is the default: at the same MFMA rate as bfloat16 it keeps the
# encoder output within 1e-2 of the fp32 reference, which bfloat16 rounding alone already exceeds.
_COMPUTE_DTYPES = {"float16": _lib.DT_F16, "fp16": _lib.DT_F16, "bfloat16": _lib.DT_BF16, "bf16": _lib.DT_BF16}
_OUT_DTYPES = {torch.float32: _lib.DT_F32, torch.bfloat16: _lib.DT_BF16, torch.float16: _lib.DT_F16}


class SANMEncoder:
    def __init__(self, input_size: int = 560, output_size: int = 512, attention_heads: int = 4,
                 linear_units: int = 2048, num_blocks: int = 50, kernel_size: int = 11, sanm_shfit: int = 0,
                 input_layer: str = "pe", normalize_before: bool = True, selfattention_layer_type: str = "sanm",
                 compute_dtype: str = "float16", ctx: _lib.Context | None = None, small_batch_rows: int | None = None,
                 **kwargs):
        # small_batch_rows: calls of at most this many rows (batch * T') take the latency launch plan (library default
        # 2048; 0 switches it off and saves the row-major weight copies it reads)
        if input_layer != "pe" or not normalize_before or selfattention_layer_type != "sanm":
            raise ValueError("SANMEncoder: only input_layer='pe', normalize_before=True, 'sanm' attention are built")
        if ctx is None:
            cfg = _lib.default_config()
            cfg.enc_input_size = input_size
            cfg.enc_output_size = output_size
            cfg.enc_heads = attention_heads
            cfg.enc_linear_units = linear_units
            cfg.enc_num_blocks = num_blocks
            cfg.enc_kernel_size = kernel_size
            cfg.enc_sanm_shift = sanm_shfit
            cfg.enc_compute_dtype = _COMPUTE_DTYPES[compute_dtype]
            if small_batch_rows is not None:
                cfg.enc_small_batch_rows = int(small_batch_rows)
            ctx = _lib.Context(cfg)
        self.ctx = ctx
        self._input_size = input_size
        self._output_size = output_size
        self.num_blocks = num_blocks
        self._workspace = None
        self._loaded = False

    def output_size(self) -> int:
        return self._output_size

    # ---- weights ---------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict: bool = True):
        """Takes a FunASR encoder ``state_dict`` (torch tensors or numpy arrays; an ``encoder.`` prefix is allowed)."""
        keep = []  # keep host buffers alive during the call
        descs = []
        for name, t in state_dict.items():
            if isinstance(t, torch.Tensor):
                a = t.detach().to("cpu", torch.float32).contiguous().numpy()
            else:
                a = np.ascontiguousarray(np.asarray(t, dtype=np.float32))
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
        self.ctx.check(self.ctx.lib.asrfe_load_encoder_weights(self.ctx.handle, arr, len(descs)))
        self._loaded = True
        return self

    # ---- forward ---------------------------------------------------------------------------
    def workspace_bytes(self, batch: int, tp: int) -> int:
        n = int(self.ctx.lib.asrfe_encoder_workspace_bytes(self.ctx.handle, batch, tp))
        if n < 0:
            raise ValueError("SANMEncoder: bad batch / length")
        return n

    def _get_workspace(self, nbytes: int, device):
        ws = self._workspace
        if ws is None or ws.numel() < nbytes or ws.device != device:
            ws = torch.empty((nbytes + 256,), dtype=torch.uint8, device=device)
            self._workspace = ws
        return ws

    def forward(self, xs_pad: torch.Tensor, ilens, prev_states=None, ctc=None, out_dtype=torch.float32,
                workspace: torch.Tensor | None = None):
        """xs_pad [B, T', input_size] float32 CUDA, ilens [B] -> (xs [B, T', output_size], olens [B] int32, None)."""
        if not xs_pad.is_cuda or xs_pad.dtype != torch.float32 or xs_pad.dim() != 3:
            raise ValueError("SANMEncoder: xs_pad must be a float32 CUDA tensor [B, T', input_size]")
        if xs_pad.shape[2] != self._input_size:
            raise ValueError(f"SANMEncoder: last dim must be {self._input_size}")
        if out_dtype not in _OUT_DTYPES:
            raise ValueError("SANMEncoder: out_dtype must be float32, bfloat16 or float16")
        xs_pad = xs_pad.contiguous()
        b, tp, _ = xs_pad.shape
        lens = torch.as_tensor(ilens)
        if lens.numel() != b:
            raise ValueError("SANMEncoder: ilens must have one entry per row")
        lens_dev = lens.to(device=xs_pad.device, dtype=torch.int32).contiguous()
        out = torch.empty((b, tp, self._output_size), dtype=out_dtype, device=xs_pad.device)
        if b and tp:
            nbytes = self.workspace_bytes(b, tp)
            ws = workspace if workspace is not None else self._get_workspace(nbytes, xs_pad.device)
            ptr = (ws.data_ptr() + 255) // 256 * 256
            avail = ws.numel() * ws.element_size() - (ptr - ws.data_ptr())
            with torch.cuda.device(xs_pad.device):
                self.ctx.check(self.ctx.lib.asrfe_encoder_forward(
                    self.ctx.handle, xs_pad.data_ptr(), lens_dev.data_ptr(), b, tp, out.data_ptr(),
                    _OUT_DTYPES[out_dtype], ptr, avail,
                    C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        olens = torch.clamp(lens_dev, min=0, max=tp)
        return out, olens, None

    __call__ = forward
