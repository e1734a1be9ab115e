"""Fixed-shape runner for the hot path: PCM batch -> fbank/LFR/CMVN -> SAN-M encoder.

Stands in for the batch forward inside the reference's
``speaker_diarization_pipeline(wav, batch_size_s=..., batch_size_token_threshold_s=40)`` call
(/root/reference/voice-service/app/services/voice_interface.py:2049-2053): the pipeline sorts VAD
segments by length, groups them into batches and runs frontend + encoder per batch.  ``AsrEngine``
is one such batch slot with every buffer allocated up front, so ``run`` only enqueues kernels on the
current stream (no allocation, no host synchronisation).

``run_ragged`` is the same step for clips of very different lengths: like the pipeline it orders the clips by
length and cuts the order into buckets, and every bucket runs at its own longest clip's frame count instead of
the batch's, so short clips are not padded to the longest one in the batch.

Multi-GPU: utterances are independent, so a batch is split into contiguous ranges, one process and one
engine per GPU (``shard_range``); nothing is exchanged between ranks.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .encoder import SANMEncoder
from .frontend import WavFrontend


def shard_range(n_items: int, rank: int, world_size: int):
    """Contiguous split of ``n_items`` utterances: rank r gets [lo, hi); sizes differ by at most one."""
    if world_size < 1 or not (0 <= rank < world_size) or n_items < 0:
        raise ValueError("shard_range: bad rank / world_size / n_items")
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_by_length(host_lens, rank: int, world_size: int):
    """Indices of the clips rank ``rank`` takes when a ragged batch is split over ``world_size`` ranks: the clips are
    ordered by length and dealt out in a serpentine (0 1 .. w-1, w-1 .. 1 0, ..), so every rank gets the same number of
    clips (+-1), nearly the same amount of audio and the same spread of lengths for its own length buckets.  A host-side
    decision; nothing is exchanged between ranks (the reference's pipeline sorts by length before batching,
    voice_interface.py:2049-2053)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("shard_by_length: bad rank / world_size")
    order = sorted(range(len(host_lens)), key=lambda i: (int(host_lens[i]), i))
    mine = []
    for pos, idx in enumerate(order):
        lap, k = divmod(pos, world_size)
        owner = k if lap % 2 == 0 else world_size - 1 - k
        if owner == rank:
            mine.append(idx)
    return mine


class AsrEngine:
    """``n_streams`` > 1 splits the batch into that many contiguous chunks, each with its own buffers and HIP
    stream: the kernels of different chunks overlap on the GPU (one chunk's memory-bound phases run beside
    another's MFMA phases), which a single in-order stream cannot do.  Results are identical to n_streams = 1."""

    def __init__(self, batch: int, max_samples: int, cfg: _lib.Config | None = None, device=None,
                 out_dtype=torch.float32, cmvn: tuple | None = None, state_dict=None, n_streams: int = 1,
                 n_buckets: int = 0, session_float: bool = False, ctx: _lib.Context | None = None):
        # session_float: float32 PCM is a streaming-session buffer in [-1, 1]; the frontend kernel then applies the
        # reference's int16 hand-off (clip, x 32767, truncate; voice_interface.py:1997-2010) while loading, so the
        # result equals that of the int16 WAV the reference writes.  Otherwise float32 PCM is scaled by 32768 as
        # FunASR's loader does.
        self.session_float = bool(session_float)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        # ctx: another engine's context on the same device (its configuration, CMVN table and encoder weights are then
        # shared: pass neither cfg, cmvn nor state_dict); engines of different batch sizes can serve one model that way
        if ctx is not None:
            if cfg is not None or cmvn is not None or state_dict is not None:
                raise ValueError("AsrEngine: a shared context brings its own configuration, CMVN table and weights")
            self.ctx = ctx
        else:
            with torch.cuda.device(self.device):
                self.ctx = _lib.Context(cfg)
        c = self.ctx.cfg
        self.frontend = WavFrontend(fs=c.sample_rate, n_mels=c.n_mels, lfr_m=c.lfr_m, lfr_n=c.lfr_n, ctx=self.ctx,
                                    cmvn=cmvn)
        self.encoder = SANMEncoder(input_size=c.enc_input_size, output_size=c.enc_output_size,
                                   attention_heads=c.enc_heads, linear_units=c.enc_linear_units,
                                   num_blocks=c.enc_num_blocks, kernel_size=c.enc_kernel_size,
                                   sanm_shfit=c.enc_sanm_shift, ctx=self.ctx)
        if state_dict is not None:
            self.encoder.load_state_dict(state_dict)
        self.batch = batch
        self.max_samples = max_samples
        self.t_max = self.frontend.num_frames(max_samples)
        self.tp_max = max(1, self.frontend.num_lfr_frames(self.t_max))
        dev = self.device
        self.feats = torch.empty((batch, self.tp_max, c.lfr_m * c.n_mels), dtype=torch.float32, device=dev)
        self.feat_lens = torch.zeros((batch,), dtype=torch.int32, device=dev)
        self.out = torch.empty((batch, self.tp_max, c.enc_output_size), dtype=out_dtype, device=dev)
        self.out_dtype = {torch.float32: _lib.DT_F32, torch.bfloat16: _lib.DT_BF16, torch.float16: _lib.DT_F16}[out_dtype]
        n_streams = max(1, min(int(n_streams), batch))
        self.chunks = [shard_range(batch, i, n_streams) for i in range(n_streams)]
        self._ws = []
        for lo, hi in self.chunks:
            nbytes = self.encoder.workspace_bytes(hi - lo, self.tp_max)
            buf = torch.empty((nbytes + 256,), dtype=torch.uint8, device=dev)
            self._ws.append((buf, (buf.data_ptr() + 255) // 256 * 256, nbytes))
        with torch.cuda.device(dev):
            self._streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)] if n_streams > 1 else [None]
            self._done = [torch.cuda.Event() for _ in range(n_streams)]
        # length buckets of run_ragged (0: that entry point is not used and nothing is allocated for it)
        n_buckets = max(0, min(int(n_buckets), batch))
        self.buckets = [shard_range(batch, i, n_buckets) for i in range(n_buckets)]
        self._bucket_ws = []
        for lo, hi in self.buckets:
            nbytes = self.encoder.workspace_bytes(hi - lo, self.tp_max)
            buf = torch.empty((nbytes + 256,), dtype=torch.uint8, device=dev)
            self._bucket_ws.append((buf, (buf.data_ptr() + 255) // 256 * 256, nbytes))
        self._pcm_sorted = None  # [batch, max_samples] in the PCM dtype of the first ragged call
        self._lens_sorted = torch.zeros((batch,), dtype=torch.int32, device=dev) if n_buckets else None
        self._order_dev = torch.zeros((batch,), dtype=torch.int64, device=dev) if n_buckets else None

    def _check_pcm(self, pcm: torch.Tensor, pcm_lens: torch.Tensor):
        if pcm.device != self.device or pcm_lens.device != self.device or pcm_lens.dtype != torch.int32:
            raise ValueError("AsrEngine: pcm and pcm_lens (int32) must live on the engine's device")
        if pcm.dim() != 2 or pcm.shape[0] != self.batch or pcm.shape[1] > self.max_samples or pcm.stride(1) != 1:
            raise ValueError("AsrEngine: pcm must be [batch, n <= max_samples] with unit inner stride")
        dt = {torch.int16: _lib.PCM_I16,
              torch.float32: _lib.PCM_F32_WIRE if self.session_float else _lib.PCM_F32}.get(pcm.dtype)
        if dt is None:
            raise ValueError("AsrEngine: PCM dtype must be int16 or float32")
        return dt

    def _frontend_chunk(self, pcm, pcm_lens, dt, lo, hi, stream):
        self.ctx.check(self.ctx.lib.asrfe_frontend(
            self.ctx.handle, pcm.data_ptr() + lo * pcm.stride(0) * pcm.element_size(), dt, pcm.stride(0),
            pcm_lens.data_ptr() + 4 * lo, hi - lo, self.tp_max, self.feats[lo:hi].data_ptr(), _lib.DT_F32,
            self.feat_lens.data_ptr() + 4 * lo, stream))

    def _encoder_chunk(self, i, lo, hi, stream):
        _, ptr, nbytes = self._ws[i]
        self.ctx.check(self.ctx.lib.asrfe_encoder_forward(
            self.ctx.handle, self.feats[lo:hi].data_ptr(), self.feat_lens.data_ptr() + 4 * lo, hi - lo, self.tp_max,
            self.out[lo:hi].data_ptr(), self.out_dtype, ptr, nbytes, stream))

    def _run(self, pcm, pcm_lens, frontend: bool, encoder: bool):
        dt = self._check_pcm(pcm, pcm_lens) if frontend else None
        cur = torch.cuda.current_stream(self.device)
        if len(self.chunks) == 1:
            st = C.c_void_p(cur.cuda_stream)
            if frontend:
                self._frontend_chunk(pcm, pcm_lens, dt, 0, self.batch, st)
            if encoder:
                self._encoder_chunk(0, 0, self.batch, st)
            return
        for i, (lo, hi) in enumerate(self.chunks):
            s = self._streams[i]
            s.wait_stream(cur)  # inputs produced on the caller's stream are ready
            st = C.c_void_p(s.cuda_stream)
            if frontend:
                self._frontend_chunk(pcm, pcm_lens, dt, lo, hi, st)
            if encoder:
                self._encoder_chunk(i, lo, hi, st)
            self._done[i].record(s)
        for ev in self._done:
            cur.wait_event(ev)

    def run_single_stream(self, pcm: torch.Tensor, pcm_lens: torch.Tensor):
        """The same step as ``run`` on the current stream only, whatever ``n_streams`` is (per-kernel timing wants
        the launches back to back on one stream).  Its full-batch workspace is allocated on first use."""
        if len(self.chunks) == 1:
            return self.run(pcm, pcm_lens)
        if not hasattr(self, "_ws_full"):
            nbytes = self.encoder.workspace_bytes(self.batch, self.tp_max)
            buf = torch.empty((nbytes + 256,), dtype=torch.uint8, device=self.device)
            self._ws_full = (buf, (buf.data_ptr() + 255) // 256 * 256, nbytes)
        dt = self._check_pcm(pcm, pcm_lens)
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        self._frontend_chunk(pcm, pcm_lens, dt, 0, self.batch, st)
        _, ptr, nbytes = self._ws_full
        self.ctx.check(self.ctx.lib.asrfe_encoder_forward(
            self.ctx.handle, self.feats.data_ptr(), self.feat_lens.data_ptr(), self.batch, self.tp_max,
            self.out.data_ptr(), self.out_dtype, ptr, nbytes, st))
        return self.out, self.feat_lens

    def run_ragged(self, pcm: torch.Tensor, pcm_lens: torch.Tensor, host_lens):
        """The step for clips of mixed lengths.  ``host_lens`` are the sample counts as the host knows them (the same
        numbers as ``pcm_lens``; reading them back from the device would stall the stream).  Clips are ordered by
        length (stable) and the order is cut into the engine's ``n_buckets`` ranges; bucket i is computed densely at
        its own frame count T'_i.  Returns ``(order, parts)``: ``order[j]`` is the index in ``pcm`` of the j-th clip
        of the ordering, ``parts[i] = (lo, hi, out, out_lens)`` with ``out`` a ``[hi - lo, T'_i, 512]`` view of the
        engine's output buffer for clips ``order[lo:hi]``.  Only device-to-device copies and kernel launches are
        enqueued; the sorted PCM staging buffer is allocated on the first call."""
        if not self.buckets:
            raise ValueError("AsrEngine.run_ragged: the engine was built with n_buckets = 0")
        dt = self._check_pcm(pcm, pcm_lens)
        hl = [int(v) for v in host_lens]
        if len(hl) != self.batch or (hl and (min(hl) < 0 or max(hl) > pcm.shape[1])):
            raise ValueError("AsrEngine.run_ragged: host_lens must hold one in-range sample count per clip")
        order = sorted(range(self.batch), key=lambda i: hl[i])
        if self._pcm_sorted is None or self._pcm_sorted.dtype != pcm.dtype:
            self._pcm_sorted = torch.zeros((self.batch, self.max_samples), dtype=pcm.dtype, device=self.device)
        ps = self._pcm_sorted
        self._order_dev.copy_(torch.tensor(order, dtype=torch.int64), non_blocking=True)
        if pcm.shape[1] == self.max_samples:
            torch.index_select(pcm, 0, self._order_dev, out=ps)
        else:  # narrower input: the gather goes through a temporary
            ps[:, :pcm.shape[1]].copy_(pcm.index_select(0, self._order_dev))
        torch.index_select(pcm_lens, 0, self._order_dev, out=self._lens_sorted)
        cur = torch.cuda.current_stream(self.device)
        c = self.ctx.cfg
        feat_dim, out_dim = c.lfr_m * c.n_mels, c.enc_output_size
        parts = []
        used = set()
        for i, (lo, hi) in enumerate(self.buckets):
            n = hi - lo
            tp = max(1, self.frontend.num_lfr_frames(self.frontend.num_frames(max(hl[j] for j in order[lo:hi]))))
            k = i % len(self._streams)
            s = self._streams[k]
            if s is None:
                st = C.c_void_p(cur.cuda_stream)
            else:
                if k not in used:
                    s.wait_stream(cur)
                st = C.c_void_p(s.cuda_stream)
            used.add(k)
            feats = self.feats[lo:hi]
            self.ctx.check(self.ctx.lib.asrfe_frontend(
                self.ctx.handle, ps.data_ptr() + lo * ps.stride(0) * ps.element_size(), dt, ps.stride(0),
                self._lens_sorted.data_ptr() + 4 * lo, n, tp, feats.data_ptr(), _lib.DT_F32,
                self.feat_lens.data_ptr() + 4 * lo, st))
            _, ptr, nbytes = self._bucket_ws[i]
            self.ctx.check(self.ctx.lib.asrfe_encoder_forward(
                self.ctx.handle, feats.data_ptr(), self.feat_lens.data_ptr() + 4 * lo, n, tp,
                self.out[lo:hi].data_ptr(), self.out_dtype, ptr, nbytes, st))
            out = self.out[lo:hi].reshape(-1)[: n * tp * out_dim].view(n, tp, out_dim)
            parts.append((lo, hi, out, self.feat_lens[lo:hi]))
        for k in used:
            if self._streams[k] is not None:
                self._done[k].record(self._streams[k])
                cur.wait_event(self._done[k])
        return order, parts

    def stage(self, pcm_host: torch.Tensor, lens_host: torch.Tensor, slot: int):
        """Starts the copy of a batch from pinned host memory into device slot 0 or 1 on the engine's copy stream
        and returns at once; ``run_staged(slot)`` runs the step on it.  Staging batch i + 1 before running batch i
        hides the PCIe transfer behind the compute of the batch in flight.  The two slots are allocated on first use."""
        if slot not in (0, 1):
            raise ValueError("AsrEngine.stage: slot must be 0 or 1")
        if pcm_host.dim() != 2 or pcm_host.shape[0] != self.batch or pcm_host.shape[1] > self.max_samples:
            raise ValueError("AsrEngine.stage: pcm_host must be [batch, n <= max_samples]")
        if not hasattr(self, "_slots") or self._slots[0][0].dtype != pcm_host.dtype:
            with torch.cuda.device(self.device):
                self._copy_stream = torch.cuda.Stream(device=self.device)
                self._slots = [(torch.zeros((self.batch, self.max_samples), dtype=pcm_host.dtype, device=self.device),
                                torch.zeros((self.batch,), dtype=torch.int32, device=self.device),
                                torch.cuda.Event(), torch.cuda.Event()) for _ in range(2)]
                # the zero fills above run on the current stream: the first copy must not overtake them
                self._copy_stream.wait_stream(torch.cuda.current_stream(self.device))
        pcm_dev, lens_dev, copied, consumed = self._slots[slot]
        cs = self._copy_stream
        cs.wait_event(consumed)  # the step that last read this slot has finished with it
        with torch.cuda.stream(cs):
            pcm_dev[:, :pcm_host.shape[1]].copy_(pcm_host, non_blocking=True)
            lens_dev.copy_(lens_host.to(torch.int32), non_blocking=True)
            copied.record(cs)

    def run_staged(self, slot: int):
        """The step on the batch ``stage`` put into ``slot``; same return value as ``run``."""
        pcm_dev, lens_dev, copied, consumed = self._slots[slot]
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(copied)
        res = self.run(pcm_dev, lens_dev)
        consumed.record(cur)
        return res

    def run_frontend(self, pcm: torch.Tensor, pcm_lens: torch.Tensor):
        """pcm [batch, >= max len] int16/float32 and pcm_lens [batch] int32, both on this device."""
        self._run(pcm, pcm_lens, True, False)
        return self.feats, self.feat_lens

    def run_encoder(self):
        self._run(None, None, False, True)
        return self.out, self.feat_lens

    def run(self, pcm: torch.Tensor, pcm_lens: torch.Tensor):
        """One step of the hot path; returns (encoder_out [batch, T', 512], out_lens [batch]) views of engine buffers."""
        self._run(pcm, pcm_lens, True, True)
        return self.out, self.feat_lens
