"""CPU oracle for the speech frontend: Kaldi-style fbank -> LFR -> CMVN.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker.

PARITY UNPINNED: the arithmetic restated here is not in the reference tree.
The reference reaches it through third-party packages that are not vendored
(``funasr`` unpinned, /root/reference/voice-service/Dockerfile:25-36;
``torchaudio.compliance.kaldi.fbank`` underneath it) at these call sites:
/root/reference/voice-service/app/services/voice_interface.py:2049-2053 (ASR),
:1585-1590 (VAD) and :1370-1374 (KWS).  The reference holds no test, fixture
or golden vector for this path, so the restatement follows the published
algorithm of Kaldi ``compute-fbank-feats`` (feature-window.cc,
feature-fbank.cc, mel-computations.cc) as wrapped by torchaudio's
``kaldi.fbank`` and FunASR's ``WavFrontend`` (``apply_lfr``, ``apply_cmvn``),
and is cross-checked against an independent implementation that is importable
here (``transformers.audio_utils.spectrogram`` in its Kaldi configuration; see
tests/golden/make_golden.py).

The PCM contract comes from the reference itself: float PCM is exactly
int16/32768 (voice_interface.py:1009-1010,1046) and the ASR pipeline is handed
a 16 kHz mono int16 WAV (voice_interface.py:1997-2023), so int16 is the
canonical input and the frontend's ``x * 32768`` up-scaling restores it.

Everything is computed in float64 so that the oracle is the "truth" both the
float32 reference stack and the float32 HIP kernels are compared with.
"""
from __future__ import annotations

import math

import numpy as np

SAMPLE_RATE = 16000
FRAME_LEN = 400  # 25 ms
FRAME_SHIFT = 160  # 10 ms
N_FFT = 512  # round_to_power_of_two(400)
N_MELS = 80
PREEMPH = 0.97
LOW_FREQ = 20.0
HIGH_FREQ = 8000.0  # torchaudio: high_freq=0 -> Nyquist
LOG_FLOOR = float(np.finfo(np.float32).eps)  # 1.1920929e-07
LFR_M = 7
LFR_N = 6


def num_frames(n_samples: int, frame_len: int = FRAME_LEN, shift: int = FRAME_SHIFT) -> int:
    """snip_edges=True frame count: frame f covers samples [shift*f, shift*f + frame_len)."""
    if n_samples < frame_len:
        return 0
    return 1 + (n_samples - frame_len) // shift


def num_lfr_frames(t: int, lfr_n: int = LFR_N) -> int:
    """FunASR apply_lfr: T' = ceil(T / n)."""
    return (t + lfr_n - 1) // lfr_n


def hamming_window(n: int = FRAME_LEN) -> np.ndarray:
    """Symmetric (non-periodic) Hamming: 0.54 - 0.46 cos(2 pi i / (n - 1))."""
    i = np.arange(n, dtype=np.float64)
    return 0.54 - 0.46 * np.cos(2.0 * np.pi * i / (n - 1))


def mel_scale(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_banks(
    n_mels: int = N_MELS,
    n_fft: int = N_FFT,
    sample_rate: int = SAMPLE_RATE,
    low_freq: float = LOW_FREQ,
    high_freq: float = HIGH_FREQ,
) -> np.ndarray:
    """Kaldi triangular filters, built in the mel domain, no area normalisation.

    Returns (n_mels, n_fft // 2 + 1); the Nyquist column is zero (Kaldi computes
    n_fft // 2 bins and torchaudio pads one zero column).
    """
    n_bins = n_fft // 2
    bin_width = sample_rate / n_fft
    mel_lo = mel_scale(low_freq)
    mel_hi = mel_scale(high_freq)
    delta = (mel_hi - mel_lo) / (n_mels + 1)
    m = np.arange(n_mels, dtype=np.float64)[:, None]
    left = mel_lo + m * delta
    center = mel_lo + (m + 1.0) * delta
    right = mel_lo + (m + 2.0) * delta
    mel = mel_scale(bin_width * np.arange(n_bins, dtype=np.float64))[None, :]
    up = (mel - left) / (center - left)
    down = (right - mel) / (right - center)
    w = np.maximum(0.0, np.minimum(up, down))
    return np.concatenate([w, np.zeros((n_mels, 1))], axis=1)


def to_kaldi_scale(pcm: np.ndarray) -> np.ndarray:
    """int16 stays as is; float PCM in [-1, 1) is multiplied by 32768 (WavFrontend upsacle_samples)."""
    pcm = np.asarray(pcm)
    if pcm.dtype == np.int16:
        return pcm.astype(np.float64)
    if pcm.dtype in (np.float32, np.float64):
        return pcm.astype(np.float64) * 32768.0
    raise TypeError(f"unsupported PCM dtype {pcm.dtype}")


def frame_signal(x: np.ndarray) -> np.ndarray:
    t = num_frames(len(x))
    if t == 0:
        return np.zeros((0, FRAME_LEN), dtype=np.float64)
    idx = FRAME_SHIFT * np.arange(t)[:, None] + np.arange(FRAME_LEN)[None, :]
    return x[idx]


def kaldi_fbank(pcm: np.ndarray, window: np.ndarray | None = None, banks: np.ndarray | None = None) -> np.ndarray:
    """One utterance -> (T, 80) float64 log-mel (dither 0, energy_floor 0, use_power, use_log_fbank)."""
    x = to_kaldi_scale(pcm)
    frames = frame_signal(x)
    if frames.shape[0] == 0:
        return np.zeros((0, N_MELS), dtype=np.float64)
    # remove_dc_offset: per-frame mean
    frames = frames - frames.mean(axis=1, keepdims=True)
    # pre-emphasis inside the frame, first sample uses itself as predecessor
    prev = np.concatenate([frames[:, :1], frames[:, :-1]], axis=1)
    frames = frames - PREEMPH * prev
    frames = frames * (hamming_window() if window is None else window)[None, :]
    spec = np.fft.rfft(frames, n=N_FFT, axis=1)
    power = spec.real**2 + spec.imag**2
    mel = power @ (mel_banks() if banks is None else banks).T
    return np.log(np.maximum(mel, LOG_FLOOR))


def apply_lfr(feats: np.ndarray, m: int = LFR_M, n: int = LFR_N) -> np.ndarray:
    """FunASR apply_lfr: left-pad (m-1)//2 copies of frame 0, stack m frames every n, tail repeats the last frame."""
    t = feats.shape[0]
    d = feats.shape[1]
    tp = num_lfr_frames(t, n)
    if t == 0:
        return np.zeros((0, m * d), dtype=feats.dtype)
    left = (m - 1) // 2
    src = np.clip(n * np.arange(tp)[:, None] + np.arange(m)[None, :] - left, 0, t - 1)
    return feats[src].reshape(tp, m * d)


def apply_cmvn(feats: np.ndarray, shift: np.ndarray, scale: np.ndarray) -> np.ndarray:
    """FunASR apply_cmvn on an am.mvn: y = (x + <AddShift>) * <Rescale>."""
    return (feats + shift[None, :]) * scale[None, :]


def frontend(pcm_batch, lens, shift=None, scale=None):
    """Batch frontend.  pcm_batch: (B, Nmax) int16 or float; lens: (B,) sample counts.

    Returns (feats (B, T'max, 560) float64 zero-padded, feats_lens (B,) int32), the contract of
    FunASR WavFrontend.forward(input, input_lengths).
    """
    outs = []
    for b in range(len(lens)):
        f = apply_lfr(kaldi_fbank(np.asarray(pcm_batch[b])[: int(lens[b])]))
        if shift is not None:
            f = apply_cmvn(f, np.asarray(shift, dtype=np.float64), np.asarray(scale, dtype=np.float64))
        outs.append(f)
    tp = max((o.shape[0] for o in outs), default=0)
    dim = LFR_M * N_MELS
    feats = np.zeros((len(outs), tp, dim), dtype=np.float64)
    for b, o in enumerate(outs):
        feats[b, : o.shape[0]] = o
    return feats, np.array([o.shape[0] for o in outs], dtype=np.int32)


def fbank_batch(pcm_batch, lens):
    """(B, Nmax) -> (logmel (B, Tmax, 80) float64 zero-padded, n_frames (B,) int32)."""
    outs = [kaldi_fbank(np.asarray(pcm_batch[b])[: int(lens[b])]) for b in range(len(lens))]
    t = max((o.shape[0] for o in outs), default=0)
    res = np.zeros((len(outs), t, N_MELS), dtype=np.float64)
    for b, o in enumerate(outs):
        res[b, : o.shape[0]] = o
    return res, np.array([o.shape[0] for o in outs], dtype=np.int32)


def load_cmvn(path: str):
    """Parse a Kaldi-nnet am.mvn (<AddShift> / <Rescale> with <LearnRateCoef> rows), as FunASR load_cmvn does."""
    with open(path, "r", encoding="utf-8") as f:
        lines = f.readlines()
    shift = scale = None
    for i, line in enumerate(lines):
        items = line.split()
        if not items:
            continue
        if items[0] == "<AddShift>":
            nxt = lines[i + 1].split()
            if nxt[0] == "<LearnRateCoef>":
                shift = np.array([float(v) for v in nxt[3 : len(nxt) - 1]], dtype=np.float64)
        elif items[0] == "<Rescale>":
            nxt = lines[i + 1].split()
            if nxt[0] == "<LearnRateCoef>":
                scale = np.array([float(v) for v in nxt[3 : len(nxt) - 1]], dtype=np.float64)
    if shift is None or scale is None:
        raise ValueError(f"{path}: no <AddShift>/<Rescale> rows")
    return shift, scale


def synth_pcm(kind: str, n: int, seed: int) -> np.ndarray:
    """Seeded int16 test signals shared by the fixtures, the tests and bench.py."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / SAMPLE_RATE
    if kind == "uniform":
        x = rng.integers(-32768, 32768, size=n).astype(np.float64)
    elif kind == "gauss":
        x = rng.normal(0.0, 3000.0, size=n)
    elif kind == "chirp":
        f0, f1 = 50.0, 7900.0
        dur = max(n / SAMPLE_RATE, 1e-9)
        x = 12000.0 * np.sin(2 * np.pi * (f0 * t + 0.5 * (f1 - f0) * t * t / dur))
    elif kind == "bursts":
        x = np.zeros(n)
        pos = 0
        while pos < n:
            gap = int(rng.integers(800, 4000))
            blen = int(rng.integers(1600, 8000))
            s, e = min(n, pos + gap), min(n, pos + gap + blen)
            if e > s:
                tt = np.arange(e - s) / SAMPLE_RATE
                f = rng.uniform(100, 3000)
                env = np.hanning(e - s)
                x[s:e] = env * (6000 * np.sin(2 * np.pi * f * tt) + rng.normal(0, 800, e - s))
            pos = e
    elif kind == "bench":
        # bench.py workload: Gaussian sigma 3000 with the first 10 % exactly zero (log floor frames)
        x = rng.normal(0.0, 3000.0, size=n)
        x[: n // 10] = 0.0
    else:
        raise ValueError(kind)
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


if __name__ == "__main__":
    assert num_frames(480000) == 2998 and num_lfr_frames(2998) == 500
    assert num_frames(399) == 0 and num_frames(400) == 1
    z = kaldi_fbank(np.zeros(16000, dtype=np.int16))
    assert np.allclose(z, math.log(LOG_FLOOR))
    print("ok", z.shape)
