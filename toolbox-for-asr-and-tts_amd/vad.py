"""Host-side mirror of FunASR's streaming FSMN voice-activity model over libasrfe.

The reference loads ``speech_fsmn_vad_zh-cn-16k-common-pytorch`` with ``funasr.AutoModel``
(/root/reference/voice-service/app/services/voice_interface.py:723-727) and calls it once per 400 ms chunk of a
streaming session: ``vad_model.generate(input=audio_np, cache=self.vad_cache, is_final=False, chunk_size=...)``
(:1585-1590), reading ``[{"value": [[beg_ms, end_ms], ...]}]`` (:1593-1600).  Inside that call FunASR runs
``WavFrontendOnline`` (fbank, LFR 5 / 1, CMVN, with the samples and frames left over from the chunk before), the ``FSMN``
encoder with one cache per memory block, and the ``E2EVadModel`` detector that turns the per-frame silence probability
into segments.  Here the features and the network run on the device through libasrfe (``asrfe_fbank``,
``asrfe_lfr_cmvn``, ``asrfe_frame_decibel``, ``asrfe_vad_forward``); the detector is a sequential state machine over one
number per 10 ms frame and stays on the host, as it does in FunASR.  There is no CPU path for the arithmetic.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np
import torch

from . import _lib
from .frontend import WavFrontend

FRAME_MS = 10
FRAME_LEN = 400
FRAME_SHIFT = 160


# ---------------------------------------------------------------------------------------------------------------------
# the network on the device
# ---------------------------------------------------------------------------------------------------------------------
class FsmnVadNet:
    """FunASR's ``FSMN`` encoder (``encoder_conf`` of the VAD model's config.yaml) on the device; takes the
    ``state_dict`` names ``encoder.*``."""

    def __init__(self, ctx: _lib.Context, input_dim: int = 400, input_affine_dim: int = 140, fsmn_layers: int = 4,
                 linear_dim: int = 250, proj_dim: int = 128, lorder: int = 20, rorder: int = 0, lstride: int = 1,
                 rstride: int = 0, output_affine_dim: int = 140, output_dim: int = 248, **kwargs):
        if rorder != 0 or lstride != 1:
            raise ValueError("FsmnVadNet: only rorder = 0 and lstride = 1 are built")
        self.ctx = ctx
        cfg = _lib.VadConfig()
        rc = ctx.lib.asrfe_default_vad_config(C.byref(cfg))
        if rc != 0:
            raise _lib.AsrfeError(rc, "asrfe_default_vad_config")
        cfg.input_dim, cfg.input_affine_dim, cfg.fsmn_layers, cfg.linear_dim = input_dim, input_affine_dim, fsmn_layers, linear_dim
        cfg.proj_dim, cfg.lorder, cfg.output_affine_dim, cfg.output_dim = proj_dim, lorder, output_affine_dim, output_dim
        self.cfg = cfg
        self.input_dim, self.output_dim = input_dim, output_dim
        self.cache_floats = fsmn_layers * (lorder - 1) * proj_dim
        self._ws = None

    def load_state_dict(self, state_dict):
        keep, descs = [], []
        for name, t in state_dict.items():
            if not name.startswith("encoder."):
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
        self.ctx.check(self.ctx.lib.asrfe_load_vad_weights(self.ctx.handle, C.byref(self.cfg), arr, len(descs)))
        return self

    def new_cache(self, batch: int, device) -> torch.Tensor:
        """Zeros: the rows in front of a stream's first frame, one row of ``cache_floats`` per stream."""
        return torch.zeros((batch, self.cache_floats), dtype=torch.float32, device=device)

    def forward(self, feats: torch.Tensor, lens=None, cache: torch.Tensor | None = None) -> torch.Tensor:
        """feats [B, T, input_dim] float32 on the device -> scores [B, T, output_dim] (softmax).  ``cache`` [B, cache_floats]
        is read as the rows before the chunk and updated in place with the chunk's last rows (``lens[b]`` of them count)."""
        if not feats.is_cuda or feats.dtype != torch.float32 or feats.dim() != 3 or feats.shape[2] != self.input_dim:
            raise ValueError(f"FsmnVadNet: feats must be a float32 CUDA tensor [B, T, {self.input_dim}]")
        x = feats.contiguous()
        b, t, _ = x.shape
        dev = x.device
        scores = torch.empty((b, t, self.output_dim), dtype=torch.float32, device=dev)
        if b == 0 or t == 0:
            return scores
        lens_dev = None
        if lens is not None:
            lens_dev = torch.as_tensor(lens).to(device=dev, dtype=torch.int32).contiguous()
            if lens_dev.numel() != b:
                raise ValueError("FsmnVadNet: one length per stream")
        if cache is not None and (not cache.is_cuda or cache.dtype != torch.float32 or not cache.is_contiguous()
                                  or cache.shape != (b, self.cache_floats)):
            raise ValueError(f"FsmnVadNet: cache must be a contiguous float32 CUDA tensor [{b}, {self.cache_floats}]")
        nbytes = int(self.ctx.lib.asrfe_vad_workspace_bytes(self.ctx.handle, b, t))
        if nbytes < 0:
            raise ValueError("FsmnVadNet: weights not loaded")
        if self._ws is None or self._ws.numel() < nbytes + 256 or self._ws.device != dev:
            self._ws = torch.empty((nbytes + 256,), dtype=torch.uint8, device=dev)
        ptr = (self._ws.data_ptr() + 255) // 256 * 256
        with torch.cuda.device(dev):
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            self.ctx.check(self.ctx.lib.asrfe_vad_forward(
                self.ctx.handle, x.data_ptr(), lens_dev.data_ptr() if lens_dev is not None else None, b, t,
                cache.data_ptr() if cache is not None else None, scores.data_ptr(), ptr,
                self._ws.numel() - (ptr - self._ws.data_ptr()), st))
        return scores

    __call__ = forward


def frame_decibel(ctx: _lib.Context, pcm: torch.Tensor, lens) -> torch.Tensor:
    """PCM [B, N] (int16, or float32 in [-1, 1]) on the device -> [B, T] 10 log10(frame energy + 1e-6)."""
    if pcm.dim() == 1:
        pcm = pcm[None]
    pcm = pcm.contiguous()
    dt = _lib.PCM_I16 if pcm.dtype == torch.int16 else _lib.PCM_F32
    if pcm.dtype not in (torch.int16, torch.float32):
        raise ValueError("frame_decibel: PCM dtype must be int16 or float32")
    lens_host = torch.as_tensor(lens).detach().to("cpu", torch.int64)
    n = int(lens_host.max()) if lens_host.numel() else 0
    t_max = 0 if n < FRAME_LEN else 1 + (n - FRAME_LEN) // FRAME_SHIFT
    out = torch.zeros((pcm.shape[0], t_max), dtype=torch.float32, device=pcm.device)
    if t_max and pcm.shape[0]:
        lens_dev = lens_host.to(torch.int32).to(pcm.device)
        with torch.cuda.device(pcm.device):
            ctx.check(ctx.lib.asrfe_frame_decibel(ctx.handle, pcm.data_ptr(), dt, pcm.stride(0), lens_dev.data_ptr(), pcm.shape[0],
                                                  t_max, out.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# the detector (host): FunASR's E2EVadModel / WindowDetector, frame by frame
# ---------------------------------------------------------------------------------------------------------------------
@dataclass
class VadOptions:
    """``model_conf`` of the VAD model's config.yaml (FunASR's VADXOptions), with that model's values."""
    sample_rate: int = 16000
    detect_mode: int = 1                  # 0 single utterance, 1 multiple utterances
    snr_mode: int = 0
    max_end_silence_time: int = 800
    max_start_silence_time: int = 3000
    do_start_point_detection: bool = True
    do_end_point_detection: bool = True
    window_size_ms: int = 200
    sil_to_speech_time_thres: int = 150
    speech_to_sil_time_thres: int = 150
    speech_2_noise_ratio: float = 1.0
    do_extend: int = 1
    lookback_time_start_point: int = 200
    lookahead_time_end_point: int = 100
    max_single_segment_time: int = 60000
    nn_eval_block_size: int = 8
    dcd_block_size: int = 4
    snr_thres: float = -100.0
    noise_frame_num_used_for_snr: int = 100
    decibel_thres: float = -100.0
    speech_noise_thres: float = 0.6
    fe_prior_thres: float = 1e-4
    silence_pdf_num: int = 1
    sil_pdf_ids: list = field(default_factory=lambda: [0])
    speech_noise_thresh_low: float = -0.1
    speech_noise_thresh_high: float = 0.3
    output_frame_probs: bool = False
    frame_in_ms: int = 10
    frame_length_ms: int = 25

    @classmethod
    def from_conf(cls, conf: dict | None) -> "VadOptions":
        known = {f for f in cls.__dataclass_fields__}
        return cls(**{k: v for k, v in (conf or {}).items() if k in known})


SIL, SPEECH = 0, 1
# what a frame does to the smoothed state
SPEECH2SPEECH, SPEECH2SIL, SIL2SIL, SIL2SPEECH = 0, 1, 2, 3
# the detector's own state
NOT_STARTED, IN_SPEECH, ENDED = 1, 2, 3


class WindowDetector:
    """A sliding window of frame decisions: the smoothed state turns to speech when at least ``sil_to_speech`` of the
    window's frames are speech and back to silence when at most ``speech_to_sil`` are."""

    def __init__(self, window_size_ms: int, sil_to_speech_time: int, speech_to_sil_time: int, frame_size_ms: int):
        self.win_size_frame = int(window_size_ms / frame_size_ms)
        self.sil_to_speech_frmcnt_thres = int(sil_to_speech_time / frame_size_ms)
        self.speech_to_sil_frmcnt_thres = int(speech_to_sil_time / frame_size_ms)
        self.reset()

    def reset(self):
        self.cur_win_pos = 0
        self.win_sum = 0
        self.win_state = [0] * self.win_size_frame
        self.pre_frame_state = SIL

    def get_win_size(self) -> int:
        return self.win_size_frame

    def detect_one_frame(self, frame_state: int) -> int:
        self.win_sum -= self.win_state[self.cur_win_pos]
        self.win_sum += frame_state
        self.win_state[self.cur_win_pos] = frame_state
        self.cur_win_pos = (self.cur_win_pos + 1) % self.win_size_frame
        if self.pre_frame_state == SIL and self.win_sum >= self.sil_to_speech_frmcnt_thres:
            self.pre_frame_state = SPEECH
            return SIL2SPEECH
        if self.pre_frame_state == SPEECH and self.win_sum <= self.speech_to_sil_frmcnt_thres:
            self.pre_frame_state = SIL
            return SPEECH2SIL
        return SIL2SIL if self.pre_frame_state == SIL else SPEECH2SPEECH


class _Segment:
    __slots__ = ("start_ms", "end_ms", "has_start", "has_end")

    def __init__(self, start_ms: int):
        self.start_ms = start_ms
        self.end_ms = start_ms
        self.has_start = False
        self.has_end = False


class VadDetector:
    """FunASR's ``E2EVadModel`` restated frame by frame: ``step(sil_prob, decibel, is_final, streaming)`` takes the new
    frames' silence probabilities (sum of the scores of ``sil_pdf_ids``) and levels and returns the segments that became
    reportable: ``[beg_ms, end_ms]``, or in streaming mode ``[beg_ms, -1]`` when a segment opens and ``[-1, end_ms]``
    when it closes."""

    def __init__(self, opts: VadOptions | None = None):
        self.opts = opts or VadOptions()
        o = self.opts
        self.windows_detector = WindowDetector(o.window_size_ms, o.sil_to_speech_time_thres, o.speech_to_sil_time_thres, o.frame_in_ms)
        self.max_end_sil_frame_cnt_thresh = o.max_end_silence_time - o.speech_to_sil_time_thres
        self.all_reset()

    # ---- state ----------------------------------------------------------------------------------------------------
    def all_reset(self):
        self.frm_cnt = 0
        self.base = 0                      # global index of sil[0] / db[0]
        self.sil: list = []
        self.db: list = []
        self.noise_average_decibel = -100.0
        self.data_buf_start_frame = 0
        self.output: list = []
        self.output_offset = 0
        self.next_seg = True
        self.number_end_time_detected = 0
        self.max_time_out = False
        self._reset_detection(first=True)

    def _reset_detection(self, first: bool = False):
        self.continous_silence_frame_count = 0
        self.latest_confirmed_speech_frame = 0
        self.lastest_confirmed_silence_frame = -1
        self.confirmed_start_frame = -1
        self.confirmed_end_frame = -1
        self.state = NOT_STARTED
        self.windows_detector.reset()
        if not first and self.output:
            # frames up to the end of the segment just closed are never looked at again
            drop = int(self.output[-1].end_ms / self.opts.frame_in_ms)
            if drop > self.base:
                del self.sil[: drop - self.base]
                del self.db[: drop - self.base]
                self.base = drop

    def _latency_frames_at_start(self) -> int:
        n = self.windows_detector.get_win_size()
        if self.opts.do_extend:
            n += int(self.opts.lookback_time_start_point / self.opts.frame_in_ms)
        return n

    # ---- bookkeeping of confirmed frames ----------------------------------------------------------------------------
    def _pop_till(self, frame_idx: int):
        if self.data_buf_start_frame < frame_idx:
            self.data_buf_start_frame = min(frame_idx, self.frm_cnt)

    def _pop_to_output(self, start_frm: int, frm_cnt: int, is_start: bool, is_end: bool):
        self._pop_till(start_frm)
        if not self.output or is_start:
            self.output.append(_Segment(start_frm * self.opts.frame_in_ms))
        seg = self.output[-1]
        self.data_buf_start_frame += frm_cnt
        seg.end_ms = (start_frm + frm_cnt) * self.opts.frame_in_ms
        if is_start:
            seg.has_start = True
        if is_end:
            seg.has_end = True

    def _on_silence_detected(self, valid_frame: int):
        self.lastest_confirmed_silence_frame = valid_frame
        if self.state == NOT_STARTED:
            self._pop_till(valid_frame)

    def _on_voice_detected(self, valid_frame: int):
        self.latest_confirmed_speech_frame = valid_frame
        self._pop_to_output(valid_frame, 1, False, False)

    def _on_voice_start(self, start_frame: int, fake_result: bool = False):
        if self.confirmed_start_frame == -1:
            self.confirmed_start_frame = start_frame
        if not fake_result and self.state == NOT_STARTED:
            self._pop_to_output(self.confirmed_start_frame, 1, True, False)

    def _on_voice_end(self, end_frame: int, fake_result: bool, is_last_frame: bool):
        for t in range(self.latest_confirmed_speech_frame + 1, end_frame):
            self._on_voice_detected(t)
        if self.confirmed_end_frame == -1:
            self.confirmed_end_frame = end_frame
        if not fake_result:
            self._pop_to_output(self.confirmed_end_frame, 1, False, True)
        self.number_end_time_detected += 1

    def _maybe_end_if_last(self, is_final_frame: bool, cur_frm_idx: int):
        if is_final_frame:
            self._on_voice_end(cur_frm_idx, False, True)
            self.state = ENDED

    # ---- one frame ----------------------------------------------------------------------------------------------------
    def _frame_state(self, t: int) -> int:
        o = self.opts
        cur_decibel = self.db[t - self.base]
        cur_snr = cur_decibel - self.noise_average_decibel
        if cur_decibel < o.decibel_thres:
            self._detect_one_frame(SIL, t, False)
            return SIL
        sil_score = min(max(self.sil[t - self.base], 1e-30), 1.0)
        noise_prob = math.log(sil_score) * o.speech_2_noise_ratio
        speech_prob = math.log(max(1.0 - sil_score, 1e-30))
        if math.exp(speech_prob) >= math.exp(noise_prob) + o.speech_noise_thres:
            return SPEECH if (cur_snr >= o.snr_thres and cur_decibel >= o.decibel_thres) else SIL
        if self.noise_average_decibel < -99.9:
            self.noise_average_decibel = cur_decibel
        else:
            n = o.noise_frame_num_used_for_snr
            self.noise_average_decibel = (cur_decibel + self.noise_average_decibel * (n - 1)) / n
        return SIL

    def _detect_one_frame(self, frame_state: int, cur_frm_idx: int, is_final_frame: bool):
        o = self.opts
        ms = o.frame_in_ms
        change = self.windows_detector.detect_one_frame(frame_state)
        too_long = lambda: cur_frm_idx - self.confirmed_start_frame + 1 > o.max_single_segment_time / ms  # noqa: E731
        if change == SIL2SPEECH:
            self.continous_silence_frame_count = 0
            if self.state == NOT_STARTED:
                start_frame = max(self.data_buf_start_frame, cur_frm_idx - self._latency_frames_at_start())
                self._on_voice_start(start_frame)
                self.state = IN_SPEECH
                for t in range(start_frame + 1, cur_frm_idx + 1):
                    self._on_voice_detected(t)
            elif self.state == IN_SPEECH:
                for t in range(self.latest_confirmed_speech_frame + 1, cur_frm_idx):
                    self._on_voice_detected(t)
                if too_long():
                    self._on_voice_end(cur_frm_idx, False, False)
                    self.state = ENDED
                elif not is_final_frame:
                    self._on_voice_detected(cur_frm_idx)
                else:
                    self._maybe_end_if_last(is_final_frame, cur_frm_idx)
        elif change in (SPEECH2SIL, SPEECH2SPEECH):
            self.continous_silence_frame_count = 0
            if self.state == IN_SPEECH:
                if too_long():
                    self.max_time_out = True
                    self._on_voice_end(cur_frm_idx, False, False)
                    self.state = ENDED
                elif not is_final_frame:
                    self._on_voice_detected(cur_frm_idx)
                else:
                    self._maybe_end_if_last(is_final_frame, cur_frm_idx)
        elif change == SIL2SIL:
            self.continous_silence_frame_count += 1
            if self.state == NOT_STARTED:
                # silence for too long in single-utterance mode, or the stream ends without any speech: an empty decision
                if (o.detect_mode == 0 and self.continous_silence_frame_count * ms > o.max_start_silence_time) or \
                        (is_final_frame and self.number_end_time_detected == 0):
                    for t in range(self.lastest_confirmed_silence_frame + 1, cur_frm_idx):
                        self._on_silence_detected(t)
                    self._on_voice_start(0, True)
                    self._on_voice_end(0, True, False)
                    self.state = ENDED
                elif cur_frm_idx >= self._latency_frames_at_start():
                    self._on_silence_detected(cur_frm_idx - self._latency_frames_at_start())
            elif self.state == IN_SPEECH:
                if self.continous_silence_frame_count * ms >= self.max_end_sil_frame_cnt_thresh:
                    lookback_frame = int(self.max_end_sil_frame_cnt_thresh / ms)
                    if o.do_extend:
                        lookback_frame -= int(o.lookahead_time_end_point / ms)
                        lookback_frame -= 1
                        lookback_frame = max(0, lookback_frame)
                    self._on_voice_end(cur_frm_idx - lookback_frame, False, False)
                    self.state = ENDED
                elif too_long():
                    self._on_voice_end(cur_frm_idx, False, False)
                    self.state = ENDED
                elif o.do_extend and not is_final_frame:
                    if self.continous_silence_frame_count <= int(o.lookahead_time_end_point / ms):
                        self._on_voice_detected(cur_frm_idx)
                else:
                    self._maybe_end_if_last(is_final_frame, cur_frm_idx)
        if self.state == ENDED and o.detect_mode == 1:
            self._reset_detection()

    # ---- one chunk ------------------------------------------------------------------------------------------------------
    def step(self, sil_prob, decibel, is_final: bool = False, streaming: bool = True) -> list:
        sil_prob = [float(v) for v in sil_prob]
        decibel = [float(v) for v in decibel]
        if len(sil_prob) != len(decibel):
            raise ValueError("VadDetector: one level per score")
        n = len(sil_prob)
        self.sil.extend(sil_prob)
        self.db.extend(decibel)
        self.frm_cnt += n
        if self.state != ENDED:
            for i in range(n - 1, -1, -1):
                t = self.frm_cnt - 1 - i
                last = is_final and i == 0
                self._detect_one_frame(self._frame_state(t), t, last)
        if is_final and n == 0 and self.state == IN_SPEECH and self.frm_cnt > 0:
            # the stream ends on a call that brings no frame: close the open segment at the last frame seen
            self._maybe_end_if_last(True, self.frm_cnt - 1)
            if self.opts.detect_mode == 1:
                self._reset_detection()
        segments = []
        for i in range(self.output_offset, len(self.output)):
            seg = self.output[i]
            if streaming:
                if not seg.has_start:
                    continue
                if not self.next_seg and not seg.has_end:
                    continue
                start_ms = seg.start_ms if self.next_seg else -1
                if seg.has_end:
                    end_ms = seg.end_ms
                    self.next_seg = True
                    self.output_offset += 1
                else:
                    end_ms = -1
                    self.next_seg = False
                segments.append([start_ms, end_ms])
            else:
                if not is_final and (not seg.has_start or not seg.has_end):
                    continue
                segments.append([seg.start_ms, seg.end_ms])
                self.output_offset += 1
        if is_final:
            self.all_reset()
        return segments


# ---------------------------------------------------------------------------------------------------------------------
# one stream's frontend state and the model object
# ---------------------------------------------------------------------------------------------------------------------
class _OnlineFrontend:
    """WavFrontendOnline for one stream on the device: the samples behind the last complete frame and the last
    ``lfr_m - 1`` log-mel frames wait for the next chunk; a feature row is emitted as soon as its right context exists,
    the rest when the stream ends (last frame repeated)."""

    def __init__(self, frontend: WavFrontend, device):
        self.fe = frontend
        self.device = device
        self.left = (frontend.lfr_m - 1) // 2
        self.tail = np.zeros((0,), dtype=np.int16)
        self.mel_tail = None        # [k, n_mels] the frames the next rows still need
        self.frames = 0             # frames seen
        self.emitted = 0            # rows emitted
        self.db_wait = np.zeros((0,), dtype=np.float32)

    def push(self, pcm: np.ndarray, is_final: bool):
        """int16 samples -> (feats [n, lfr_m * n_mels] on the device, levels [n] numpy)."""
        fe = self.fe
        x = np.concatenate([self.tail, pcm]) if self.tail.size else pcm
        n = 0 if x.shape[0] < FRAME_LEN else 1 + (x.shape[0] - FRAME_LEN) // FRAME_SHIFT
        dim = fe.output_size()
        if n:
            used = x[: FRAME_SHIFT * (n - 1) + FRAME_LEN]
            dev_pcm = torch.from_numpy(np.ascontiguousarray(used)).to(self.device)[None]
            mel, _ = fe.forward_fbank(dev_pcm, [used.shape[0]])
            db = frame_decibel(fe.ctx, dev_pcm, [used.shape[0]])[0].cpu().numpy()
            self.db_wait = np.concatenate([self.db_wait, db])
            mel = mel[0]
            self.tail = x[FRAME_SHIFT * n:]
        else:
            self.tail = x
            mel = torch.zeros((0, fe.n_mels), dtype=torch.float32, device=self.device)
        window = mel if self.mel_tail is None else torch.cat([self.mel_tail, mel], dim=0)
        first = self.frames - (0 if self.mel_tail is None else self.mel_tail.shape[0])  # global index of window[0]
        self.frames += n
        avail = self.frames if is_final else max(0, self.frames - self.left)            # rows that can exist now
        n_out = avail - self.emitted
        if n_out <= 0 or window.shape[0] == 0:
            self.mel_tail = window
            return torch.zeros((0, dim), dtype=torch.float32, device=self.device), np.zeros((0,), dtype=np.float32)
        # rows of the window: row r = frames r - left .. r + left of the window, ends repeated.  The repeats are what the
        # stream's first rows (window starts at frame 0) and, at the end of the stream, its last rows are defined to hold;
        # every other emitted row has its whole context inside the window
        rows, _ = fe.forward_lfr_cmvn(window[None], [window.shape[0]])
        lo = self.emitted - first
        feats = rows[0, lo:lo + n_out].contiguous()
        self.emitted += n_out
        keep_from = max(0, self.emitted - self.left) - first
        self.mel_tail = window[keep_from:]
        db = self.db_wait[:n_out]
        self.db_wait = self.db_wait[n_out:]
        return feats, db


class FsmnVadModel:
    """The object ``AutoModel(model=<fsmn-vad directory>)`` gives: ``generate(input=..., cache=..., is_final=...,
    chunk_size=...)`` -> ``[{"key": ..., "value": [[beg_ms, end_ms], ...]}]``."""

    def __init__(self, config: dict, cmvn, state_dict, device="cuda:0", compute_dtype: int | None = None):
        self.device = torch.device(device)
        fe_conf = dict(config.get("frontend_conf") or {})
        fe_conf.pop("cmvn_file", None)
        if float(fe_conf.get("dither", 0.0)) != 0.0:
            fe_conf["dither"] = 0.0   # FunASR's inference runs this frontend without dither as well
        fe_conf.setdefault("lfr_m", 5)
        fe_conf.setdefault("lfr_n", 1)
        if int(fe_conf["lfr_n"]) != 1:
            raise ValueError("FsmnVadModel: lfr_n = 1 only (one score per 10 ms frame)")
        with torch.cuda.device(self.device):
            cfg = _lib.default_config()
            cfg.lfr_m, cfg.lfr_n = int(fe_conf["lfr_m"]), 1
            cfg.n_mels = int(fe_conf.get("n_mels", 80))
            cfg.enc_input_size = cfg.lfr_m * cfg.n_mels
            if compute_dtype is not None:
                cfg.enc_compute_dtype = compute_dtype
            self.ctx = _lib.Context(cfg)
            self.frontend = WavFrontend(cmvn=cmvn, ctx=self.ctx, fs=int(fe_conf.get("fs", 16000)),
                                        window=str(fe_conf.get("window", "hamming")), n_mels=cfg.n_mels, lfr_m=cfg.lfr_m, lfr_n=1)
            enc = dict(config.get("encoder_conf") or {})
            enc.setdefault("input_dim", cfg.lfr_m * cfg.n_mels)
            self.net = FsmnVadNet(self.ctx, **enc).load_state_dict(state_dict)
        self.opts = VadOptions.from_conf(config.get("model_conf"))
        self.kwargs = config

    # ---- scores for a whole clip (no streaming state) ----------------------------------------------------------------------
    def frame_scores(self, pcm_int16: np.ndarray):
        """-> (scores [T, output_dim] on the device, levels [T] numpy) of one clip taken as a complete stream."""
        st = _OnlineFrontend(self.frontend, self.device)
        feats, db = st.push(np.asarray(pcm_int16, dtype=np.int16), True)
        if feats.shape[0] == 0:
            return torch.zeros((0, self.net.output_dim), device=self.device), db
        return self.net(feats[None])[0], db

    # ---- generate ----------------------------------------------------------------------------------------------------------
    @staticmethod
    def _to_int16(x) -> np.ndarray:
        from .model_dir import AutoModel
        return AutoModel._to_int16(x)

    def generate(self, input, cache: dict | None = None, is_final: bool | None = None, chunk_size: int = 60000,
                 is_streaming_input: bool | None = None, key=None, **kwargs) -> list:
        items = list(input) if isinstance(input, (list, tuple)) else [input]
        if cache is not None and len(items) != 1:
            raise ValueError("FsmnVadModel: a stream cache belongs to one input")
        results = []
        for i, item in enumerate(items):
            k = key[i] if isinstance(key, (list, tuple)) else (key if key is not None else f"rand_key_{i}")
            value = self._infer_one(self._to_int16(item), cache if cache is not None else {}, is_final, int(chunk_size),
                                    is_streaming_input)
            results.append({"key": k, "value": value})
        return results

    def _infer_one(self, pcm: np.ndarray, cache: dict, is_final, chunk_size: int, is_streaming_input) -> list:
        streaming = (chunk_size < 15000) if is_streaming_input is None else bool(is_streaming_input)
        final = (not streaming) if is_final is None else bool(is_final)
        st = cache.get("_stream")
        if st is None:
            st = cache["_stream"] = {
                "prev_samples": np.zeros((0,), dtype=np.int16),
                "frontend": _OnlineFrontend(self.frontend, self.device),
                "fsmn": self.net.new_cache(1, self.device),
                "detector": VadDetector(self.opts),
            }
        sil_ids = list(self.opts.sil_pdf_ids)
        stride = int(chunk_size * self.frontend.fs / 1000)
        audio = np.concatenate([st["prev_samples"], pcm]) if st["prev_samples"].size else pcm
        n = audio.shape[0] // stride + int(final)
        m = (audio.shape[0] % stride) * (1 - int(final))
        segments = []
        with torch.cuda.device(self.device):
            for i in range(n):
                fin_i = final and i == n - 1
                feats, db = st["frontend"].push(audio[i * stride:(i + 1) * stride], fin_i)
                if feats.shape[0]:
                    scores = self.net(feats[None], cache=st["fsmn"])[0]
                    sil = scores[:, sil_ids].sum(dim=1).cpu().numpy()
                else:
                    sil = np.zeros((0,), dtype=np.float32)
                segments.extend(st["detector"].step(sil, db, fin_i, streaming))
        # (the samples behind the last whole chunk wait for the next call)
        st["prev_samples"] = audio[audio.shape[0] - m:] if m else np.zeros((0,), dtype=np.int16)
        if final:
            cache.pop("_stream", None)
        return segments
