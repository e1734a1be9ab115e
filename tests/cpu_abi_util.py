"""ctypes access to oracle/_build/libasrfe_cpu.so, the C++ CPU restatement of the frontend + encoder entry points of
include/asrfe.h (test infrastructure; host pointers).  Used by the CPU tests and by bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_build", "libasrfe_cpu.so")


class Config(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("frame_length", C.c_int32), ("frame_shift", C.c_int32), ("n_fft", C.c_int32),
                ("n_mels", C.c_int32), ("preemph", C.c_float), ("low_freq", C.c_float), ("high_freq", C.c_float),
                ("window_type", C.c_int32), ("lfr_m", C.c_int32), ("lfr_n", C.c_int32), ("enc_input_size", C.c_int32),
                ("enc_output_size", C.c_int32), ("enc_heads", C.c_int32), ("enc_linear_units", C.c_int32),
                ("enc_num_blocks", C.c_int32), ("enc_kernel_size", C.c_int32), ("enc_sanm_shift", C.c_int32),
                ("enc_ln_eps", C.c_float), ("enc_compute_dtype", C.c_int32), ("enc_small_batch_rows", C.c_int32)]


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data_host", C.POINTER(C.c_float)), ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


_lib = None


def load(build: bool = True):
    """The library; built with oracle/Makefile first if it is missing (g++ only, a few seconds)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) and build:
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True, capture_output=True)
        lib = C.CDLL(LIB)
        lib.asrfe_last_error.restype = C.c_char_p
        lib.asrfe_version.restype = C.c_char_p
        lib.asrfe_encoder_workspace_bytes.restype = C.c_int64
        _lib = lib
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class CpuModel:
    """One context of the CPU library: frontend(pcm int16 [B, N], lens) and encoder(feats, lens)."""

    def __init__(self, num_blocks=50, state_dict=None, cmvn=None, threads=0, **cfg_over):
        self.lib = load()
        cfg = Config()
        assert self.lib.asrfe_default_config(C.byref(cfg)) == 0
        cfg.enc_num_blocks = num_blocks
        for k, v in cfg_over.items():
            setattr(cfg, k, v)
        self.cfg = cfg
        self.ctx = C.c_void_p()
        self.check(self.lib.asrfe_create(C.byref(cfg), C.byref(self.ctx)))
        self.lib.asrfe_cpu_set_threads(int(threads))
        if cmvn is not None:
            sh = np.ascontiguousarray(cmvn[0], dtype=np.float32)
            sc = np.ascontiguousarray(cmvn[1], dtype=np.float32)
            self.check(self.lib.asrfe_load_cmvn(self.ctx, _fp(sh), _fp(sc), sh.shape[0]))
        if state_dict is not None:
            keep, descs = [], []
            for name, t in state_dict.items():
                a = np.ascontiguousarray(t.detach().cpu().float().numpy() if hasattr(t, "detach") else np.asarray(t, dtype=np.float32))
                keep.append(a)
                d = TensorDesc(name.encode(), _fp(a), a.ndim)
                for i, s in enumerate(a.shape):
                    d.shape[i] = s
                descs.append(d)
            arr = (TensorDesc * len(descs))(*descs)
            self.check(self.lib.asrfe_load_encoder_weights(self.ctx, arr, len(descs)))

    def check(self, rc):
        if rc != 0:
            raise RuntimeError(f"libasrfe_cpu: status {rc}: {(self.lib.asrfe_last_error(self.ctx) or b'').decode()}")

    def set_threads(self, n):
        self.lib.asrfe_cpu_set_threads(int(n))

    def fbank(self, pcm, lens, t_max=None):
        pcm = np.ascontiguousarray(pcm)
        dt = {np.dtype(np.int16): 0, np.dtype(np.float32): 1}[pcm.dtype]
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        b = pcm.shape[0]
        if t_max is None:
            t_max = max([int(self.lib.asrfe_num_frames(self.ctx, C.c_int64(int(n)))) for n in lens] + [0])
        out = np.empty((b, t_max, self.cfg.n_mels), dtype=np.float32)
        nf = np.zeros((b,), dtype=np.int32)
        self.check(self.lib.asrfe_fbank(self.ctx, pcm.ctypes.data_as(C.c_void_p), dt, C.c_int64(pcm.shape[1]), _ip(lens), b, t_max,
                                        _fp(out), _ip(nf), None))
        return out, nf

    def frontend(self, pcm, lens, tp_max=None, pcm_dtype=None):
        pcm = np.ascontiguousarray(pcm)
        dt = pcm_dtype if pcm_dtype is not None else {np.dtype(np.int16): 0, np.dtype(np.float32): 1}[pcm.dtype]
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        b = pcm.shape[0]
        if tp_max is None:
            tp_max = max([int(self.lib.asrfe_num_lfr_frames(self.ctx, self.lib.asrfe_num_frames(self.ctx, C.c_int64(int(n))))) for n in lens] + [0])
        dim = self.cfg.lfr_m * self.cfg.n_mels
        out = np.empty((b, tp_max, dim), dtype=np.float32)
        ol = np.zeros((b,), dtype=np.int32)
        self.check(self.lib.asrfe_frontend(self.ctx, pcm.ctypes.data_as(C.c_void_p), dt, C.c_int64(pcm.shape[1]), _ip(lens), b, tp_max,
                                           out.ctypes.data_as(C.c_void_p), 0, _ip(ol), None))
        return out, ol

    def encoder(self, feats, lens):
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        b, tp, _ = feats.shape
        out = np.empty((b, tp, self.cfg.enc_output_size), dtype=np.float32)
        self.check(self.lib.asrfe_encoder_forward(self.ctx, _fp(feats), _ip(lens), b, tp, out.ctypes.data_as(C.c_void_p), 0, None,
                                                  C.c_int64(0), None))
        return out

    def close(self):
        if self.ctx:
            self.lib.asrfe_destroy(self.ctx)
            self.ctx = C.c_void_p()
