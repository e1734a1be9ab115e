"""Host-side mirror of the step behind the encoder: FunASR's ``CifPredictorV2`` and ``ParaformerSANMDecoder`` over libasrfe.

The reference gets recognised text from the ModelScope pipeline call at
/root/reference/voice-service/app/services/voice_interface.py:2049-2053 and consumes it at :2056-2067 and :2108-2181;
inside that call FunASR's ``Paraformer.inference`` runs ``predictor(encoder_out, None, mask)`` ->
``decoder(encoder_out, encoder_out_lens, pre_acoustic_embeds, pre_token_length)`` -> arg-max per token -> token list ->
text.  ``ParaformerHead`` is those steps on the device: it shares the encoder's context (one library context per
device), takes FunASR's ``state_dict`` names (``predictor.*``, ``decoder.*``) and returns token ids; characters need
the model directory's token list (``tokens.json``), which is passed in by the caller.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

SOS, EOS, BLANK = 1, 2, 0
FRAME_MS = 60  # one encoder frame: lfr_n (6) frames of 10 ms


class ParaformerHead:
    def __init__(self, ctx: _lib.Context, num_blocks: int = 16, linear_units: int = 2048, vocab_size: int = 8404,
                 kernel_size: int = 11, tail_threshold: float = 0.45, threshold: float = 1.0, smooth_factor: float = 1.0,
                 noise_threshold: float = 0.0, token_list=None):
        self.ctx = ctx
        cfg = _lib.DecoderConfig()
        rc = ctx.lib.asrfe_default_decoder_config(C.byref(cfg))
        if rc != 0:
            raise _lib.AsrfeError(rc, "asrfe_default_decoder_config")
        cfg.num_blocks, cfg.linear_units, cfg.vocab_size, cfg.kernel_size = num_blocks, linear_units, vocab_size, kernel_size
        cfg.tail_threshold, cfg.threshold = tail_threshold, threshold
        cfg.smooth_factor, cfg.noise_threshold = smooth_factor, noise_threshold
        self.cfg = cfg
        self.vocab_size = vocab_size
        self.token_list = list(token_list) if token_list is not None else None
        self._ws = None
        self._loaded = False

    # ---- weights ---------------------------------------------------------------------------
    def load_state_dict(self, state_dict):
        """Takes the ``predictor.*`` and ``decoder.*`` tensors of a FunASR Paraformer ``state_dict`` (other keys are
        ignored, so the whole model's dict may be passed)."""
        keep, descs = [], []
        for name, t in state_dict.items():
            if not (name.startswith("predictor.") or name.startswith("decoder.")):
                continue
            a = t.detach().to("cpu", torch.float32).contiguous().numpy() if isinstance(t, torch.Tensor) \
                else np.ascontiguousarray(np.asarray(t, dtype=np.float32))
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
        self.ctx.check(self.ctx.lib.asrfe_load_decoder_weights(self.ctx.handle, C.byref(self.cfg), arr, len(descs)))
        self.logits_stride = int(self.ctx.lib.asrfe_decoder_logits_stride(self.ctx.handle))
        self._loaded = True
        return self

    # ---- forward ---------------------------------------------------------------------------
    def workspace_bytes(self, batch: int, tp: int, n_max: int) -> int:
        n = int(self.ctx.lib.asrfe_decoder_workspace_bytes(self.ctx.handle, batch, tp, n_max))
        if n < 0:
            raise ValueError("ParaformerHead: weights not loaded or bad sizes")
        return n

    def _workspace(self, nbytes, device):
        if self._ws is None or self._ws.numel() < nbytes + 256 or self._ws.device != device:
            self._ws = torch.empty((nbytes + 256,), dtype=torch.uint8, device=device)
        ptr = (self._ws.data_ptr() + 255) // 256 * 256
        return ptr, self._ws.numel() - (ptr - self._ws.data_ptr())

    def forward(self, encoder_out: torch.Tensor, encoder_out_lens, n_max: int | None = None, return_logits: bool = False,
                return_alphas: bool = False, return_fire_frames: bool = False):
        """encoder_out [B, T', 512] float32 on the device, lens [B] -> dict with ``token_ids`` [B, n_max] int32 (-1 behind
        an utterance's tokens), ``token_num`` [B] int32, ``embeds`` [B, n_max, 512], and on request ``logits``
        [B, n_max, vocab], ``alphas`` [B, T' + 1] and ``fire_frames`` [B, n_max] int32 (the encoder frame at which token n
        fires, -1 behind the fires: token n spans (fire[n - 1], fire[n]] frames of 60 ms).  ``n_max`` defaults to T' + 1,
        which always suffices (every frame fires at most once)."""
        if not encoder_out.is_cuda or encoder_out.dtype != torch.float32 or encoder_out.dim() != 3 or encoder_out.shape[2] != 512:
            raise ValueError("ParaformerHead: encoder_out must be a float32 CUDA tensor [B, T', 512]")
        x = encoder_out.contiguous()
        b, tp, _ = x.shape
        dev = x.device
        n_max = tp + 1 if n_max is None else int(n_max)
        lens = torch.as_tensor(encoder_out_lens).to(device=dev, dtype=torch.int32).contiguous()
        if lens.numel() != b:
            raise ValueError("ParaformerHead: one length per utterance")
        embeds = torch.empty((b, n_max, 512), dtype=torch.float32, device=dev)
        token_num = torch.zeros((b,), dtype=torch.int32, device=dev)
        ids = torch.full((b, n_max), -1, dtype=torch.int32, device=dev)
        alphas = torch.empty((b, tp + 1), dtype=torch.float32, device=dev) if (return_alphas or return_fire_frames) else None
        fires = torch.full((b, n_max), -1, dtype=torch.int32, device=dev) if return_fire_frames else None
        logits = torch.empty((b * n_max, self.logits_stride), dtype=torch.float32, device=dev) if return_logits else None
        out = {"token_ids": ids, "token_num": token_num, "embeds": embeds}
        if b and tp:
            ptr, avail = self._workspace(self.workspace_bytes(b, tp, n_max), dev)
            with torch.cuda.device(dev):
                st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                self.ctx.check(self.ctx.lib.asrfe_cif_predictor(
                    self.ctx.handle, x.data_ptr(), lens.data_ptr(), b, tp, n_max, embeds.data_ptr(), token_num.data_ptr(),
                    alphas.data_ptr() if alphas is not None else None, ptr, avail, st))
                if fires is not None:
                    self.ctx.check(self.ctx.lib.asrfe_cif_fire_frames(self.ctx.handle, alphas.data_ptr(), b, tp, n_max,
                                                                      fires.data_ptr(), st))
                self.ctx.check(self.ctx.lib.asrfe_decoder_forward(
                    self.ctx.handle, x.data_ptr(), lens.data_ptr(), b, tp, embeds.data_ptr(), token_num.data_ptr(), n_max,
                    ids.data_ptr(), logits.data_ptr() if logits is not None else None, ptr, avail, st))
        if logits is not None:
            out["logits"] = logits.view(b, n_max, self.logits_stride)[:, :, : self.vocab_size]
        if alphas is not None and return_alphas:
            out["alphas"] = alphas
        if fires is not None:
            out["fire_frames"] = fires
        return out

    __call__ = forward

    # ---- tokens -> text ----------------------------------------------------------------------
    @staticmethod
    def clean_ids(row, vocab_size: int | None = None) -> list:
        """One utterance's ids with the padding (-1), sos, eos and blank entries dropped, as Paraformer.inference does
        (and, given the vocabulary size, anything that is not a vocabulary index)."""
        return [int(t) for t in row if int(t) >= 0 and int(t) not in (SOS, EOS, BLANK) and (vocab_size is None or int(t) < vocab_size)]

    @staticmethod
    def token_spans(row, fire_row, vocab_size: int | None = None) -> list:
        """[start_ms, end_ms] of the tokens clean_ids keeps, from the frames at which the predictor fired: token n spans
        (fire[n - 1], fire[n]] encoder frames of 60 ms (FunASR reports such per-token pairs as ``timestamp``)."""
        spans, prev = [], 0
        for t, f in zip(row, fire_row):
            t, f = int(t), int(f)
            if f < 0:
                break
            end = (f + 1) * FRAME_MS
            if t >= 0 and t not in (SOS, EOS, BLANK) and (vocab_size is None or t < vocab_size):
                spans.append([prev, end])
            prev = end
        return spans

    def ids2text(self, ids) -> str:
        """Token ids -> text with the model directory's token list: tokens are joined; a word piece ending in "@@"
        continues into the next token, other Latin-script tokens are separated by one space (FunASR's
        ``sentence_postprocess`` for Chinese / English mixed output)."""
        if self.token_list is None:
            raise ValueError("ParaformerHead: no token list was given (tokens.json of the model directory)")
        out, glue = [], False
        for t in ids:
            tok = self.token_list[t]
            if tok in ("<s>", "</s>", "<blank>", "<unk>"):
                continue
            ascii_tok = tok.isascii()
            piece = tok[:-2] if tok.endswith("@@") else tok
            if out and ascii_tok and not glue and out[-1][-1:].isascii() and out[-1][-1:] != " ":
                out.append(" ")
            out.append(piece)
            glue = tok.endswith("@@")
        return "".join(out)

    def recognize(self, encoder_out, encoder_out_lens, n_max: int | None = None) -> list:
        """-> one entry per utterance: text when a token list is known, else the list of token ids."""
        r = self.forward(encoder_out, encoder_out_lens, n_max)
        rows = r["token_ids"].cpu().tolist()
        cleaned = [self.clean_ids(row, self.vocab_size) for row in rows]
        return [self.ids2text(c) for c in cleaned] if self.token_list is not None else cleaned
