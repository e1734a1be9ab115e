"""ctypes binding of libasrfe.so (include/asrfe.h).

There is no CPU fallback: if the shared library is missing or a call fails, an exception is
raised.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C toolbox-for-asr-and-tts_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libasrfe.so")

PCM_I16, PCM_F32, PCM_F32_WIRE = 0, 1, 2
DT_F32, DT_BF16, DT_F16 = 0, 1, 2

ERROR_NAMES = {
    0: "OK",
    -1: "BAD_ARG",
    -2: "SHAPE",
    -3: "HIP_ERROR",
    -4: "UNSUPPORTED_ARCH",
    -5: "NOT_LOADED",
    -6: "WORKSPACE",
    -7: "RANGE",
}


class AsrfeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libasrfe error {code} ({ERROR_NAMES.get(code, '?')}): {msg}")
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_int32),
        ("frame_length", C.c_int32),
        ("frame_shift", C.c_int32),
        ("n_fft", C.c_int32),
        ("n_mels", C.c_int32),
        ("preemph", C.c_float),
        ("low_freq", C.c_float),
        ("high_freq", C.c_float),
        ("window_type", C.c_int32),
        ("lfr_m", C.c_int32),
        ("lfr_n", C.c_int32),
        ("enc_input_size", C.c_int32),
        ("enc_output_size", C.c_int32),
        ("enc_heads", C.c_int32),
        ("enc_linear_units", C.c_int32),
        ("enc_num_blocks", C.c_int32),
        ("enc_kernel_size", C.c_int32),
        ("enc_sanm_shift", C.c_int32),
        ("enc_ln_eps", C.c_float),
        ("enc_compute_dtype", C.c_int32),
        ("enc_small_batch_rows", C.c_int32),
    ]


class DecoderConfig(C.Structure):
    _fields_ = [
        ("num_blocks", C.c_int32),
        ("linear_units", C.c_int32),
        ("kernel_size", C.c_int32),
        ("vocab_size", C.c_int32),
        ("ln_eps", C.c_float),
        ("threshold", C.c_float),
        ("tail_threshold", C.c_float),
        ("smooth_factor", C.c_float),
        ("noise_threshold", C.c_float),
    ]


class VocoderConfig(C.Structure):
    _fields_ = [
        ("n_mels", C.c_int32),
        ("initial_channel", C.c_int32),
        ("n_upsamples", C.c_int32),
        ("upsample_rates", C.c_int32 * 8),
        ("upsample_kernels", C.c_int32 * 8),
        ("n_resblocks", C.c_int32),
        ("resblock_kernels", C.c_int32 * 4),
        ("resblock_dilations", (C.c_int32 * 3) * 4),
        ("lrelu_slope", C.c_float),
        ("final_slope", C.c_float),
    ]


class VadConfig(C.Structure):
    _fields_ = [
        ("input_dim", C.c_int32),
        ("input_affine_dim", C.c_int32),
        ("fsmn_layers", C.c_int32),
        ("linear_dim", C.c_int32),
        ("proj_dim", C.c_int32),
        ("lorder", C.c_int32),
        ("output_affine_dim", C.c_int32),
        ("output_dim", C.c_int32),
    ]


PROF_KINDS = ["fbank", "frontend", "lfr_cmvn", "pe_table", "layernorm", "gemm_qkv", "fsmn", "attention", "gemm_out",
              "gemm_ffn1", "gemm_ffn2", "post_attn"]


class Profile(C.Structure):
    _fields_ = [("launches", C.c_int32 * 12), ("total_ms", C.c_float * 12)]


class TensorDesc(C.Structure):
    _fields_ = [
        ("name", C.c_char_p),
        ("data_host", C.POINTER(C.c_float)),
        ("ndim", C.c_int32),
        ("shape", C.c_int64 * 4),
    ]


# name -> (restype, argtypes); must list every function include/asrfe.h declares
SIGNATURES = {
    "asrfe_default_config": (C.c_int32, [C.POINTER(Config)]),
    "asrfe_create": (C.c_int32, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "asrfe_destroy": (C.c_int32, [C.c_void_p]),
    "asrfe_load_cmvn": (C.c_int32, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32]),
    "asrfe_load_encoder_weights": (C.c_int32, [C.c_void_p, C.POINTER(TensorDesc), C.c_int32]),
    "asrfe_num_frames": (C.c_int32, [C.c_void_p, C.c_int64]),
    "asrfe_num_lfr_frames": (C.c_int32, [C.c_void_p, C.c_int32]),
    "asrfe_fbank": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "asrfe_lfr_cmvn": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                   C.c_int32, C.c_void_p, C.c_void_p]),
    "asrfe_frontend": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
                                   C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "asrfe_encoder_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32]),
    "asrfe_encoder_forward": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                          C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "asrfe_encoder_forward_static": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                                 C.c_int32, C.c_void_p]),
    "asrfe_default_decoder_config": (C.c_int32, [C.POINTER(DecoderConfig)]),
    "asrfe_load_decoder_weights": (C.c_int32, [C.c_void_p, C.POINTER(DecoderConfig), C.POINTER(TensorDesc), C.c_int32]),
    "asrfe_decoder_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "asrfe_decoder_logits_stride": (C.c_int32, [C.c_void_p]),
    "asrfe_cif_predictor": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "asrfe_cif_fire_frames": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "asrfe_decoder_forward": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "asrfe_default_vocoder_config": (C.c_int32, [C.POINTER(VocoderConfig)]),
    "asrfe_load_vocoder_weights": (C.c_int32, [C.c_void_p, C.POINTER(VocoderConfig), C.POINTER(TensorDesc), C.c_int32]),
    "asrfe_vocoder_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32]),
    "asrfe_vocoder_hop": (C.c_int32, [C.c_void_p]),
    "asrfe_vocoder_forward": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                          C.c_void_p]),
    "asrfe_default_vad_config": (C.c_int32, [C.POINTER(VadConfig)]),
    "asrfe_load_vad_weights": (C.c_int32, [C.c_void_p, C.POINTER(VadConfig), C.POINTER(TensorDesc), C.c_int32]),
    "asrfe_vad_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int32, C.c_int32]),
    "asrfe_vad_cache_floats": (C.c_int64, [C.c_void_p]),
    "asrfe_vad_forward": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int64, C.c_void_p]),
    "asrfe_frame_decibel": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_void_p]),
    "asrfe_host_mel_banks": (C.c_int32, [C.POINTER(Config), C.POINTER(C.c_double)]),
    "asrfe_host_window": (C.c_int32, [C.POINTER(Config), C.POINTER(C.c_double)]),
    "asrfe_profile_begin": (C.c_int32, [C.c_void_p, C.c_int32]),
    "asrfe_profile_end": (C.c_int32, [C.c_void_p, C.POINTER(Profile)]),
    "asrfe_debug_gemm": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                     C.c_int32, C.c_void_p]),
    "asrfe_debug_post_attn": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "asrfe_debug_attention": (C.c_int32, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                          C.c_void_p, C.c_int32, C.c_void_p]),
    "asrfe_last_error": (C.c_char_p, [C.c_void_p]),
    "asrfe_version": (C.c_char_p, []),
}

_lib = None


def load() -> C.CDLL:
    """Loads libasrfe.so once; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built and there is no fallback path "
                "(run __graft_entry__.build() or make -C toolbox-for-asr-and-tts_amd/csrc)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def default_config() -> Config:
    cfg = Config()
    rc = load().asrfe_default_config(C.byref(cfg))
    if rc != 0:
        raise AsrfeError(rc, "asrfe_default_config")
    return cfg


class Context:
    """Owns one asrfe_ctx on the current HIP device."""

    def __init__(self, cfg: Config | None = None):
        self.lib = load()
        self.cfg = cfg if cfg is not None else default_config()
        h = C.c_void_p()
        rc = self.lib.asrfe_create(C.byref(self.cfg), C.byref(h))
        if rc != 0:
            raise AsrfeError(rc, (self.lib.asrfe_last_error(None) or b"").decode())
        self.handle = h

    def check(self, rc: int):
        if rc != 0:
            raise AsrfeError(rc, (self.lib.asrfe_last_error(self.handle) or b"").decode())

    def profile_begin(self, max_launches: int = 4096):
        self.check(self.lib.asrfe_profile_begin(self.handle, max_launches))

    def profile_end(self) -> dict:
        """Returns {kind: (launches, total_ms)} for the launches since profile_begin (synchronises)."""
        p = Profile()
        self.check(self.lib.asrfe_profile_end(self.handle, C.byref(p)))
        return {k: (int(p.launches[i]), float(p.total_ms[i])) for i, k in enumerate(PROF_KINDS) if p.launches[i]}

    def close(self):
        if getattr(self, "handle", None):
            self.lib.asrfe_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
