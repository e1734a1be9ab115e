"""MI355X-native speech frontend + Paraformer SAN-M encoder (HIP kernels behind the libasrfe C ABI).

Only the batched hot path lives here: ``WavFrontend`` (Kaldi fbank -> LFR -> CMVN) and ``SANMEncoder``,
mirroring the FunASR classes the reference service reaches through ``funasr.AutoModel`` / the
ModelScope pipeline, plus ``AsrEngine`` which runs both back to back on preallocated buffers.
"""
from ._lib import AsrfeError, Config, Context, default_config, load  # noqa: F401
from .frontend import WavFrontend, load_cmvn  # noqa: F401
from .encoder import SANMEncoder  # noqa: F401
from .engine import AsrEngine, shard_by_length, shard_range  # noqa: F401
from .decoder import ParaformerHead  # noqa: F401
from .model_dir import AutoModel, read_model_dir  # noqa: F401
from .vad import FsmnVadModel, FsmnVadNet, VadDetector, VadOptions  # noqa: F401
from .ingress import PcmAccumulator, StreamingPcmSession, decode_wav_base64, wire_int16  # noqa: F401

__all__ = ["AsrfeError", "Config", "Context", "default_config", "load", "WavFrontend", "load_cmvn",
           "SANMEncoder", "ParaformerHead", "AutoModel", "read_model_dir", "FsmnVadModel", "FsmnVadNet", "VadDetector", "VadOptions", "AsrEngine", "shard_range", "shard_by_length", "PcmAccumulator", "StreamingPcmSession", "decode_wav_base64", "wire_int16"]
