#!/usr/bin/env python3
"""Times the fbank kernel (256 x 30 s int16 -> fp32 log-mel) and the fused frontend, and checks both against the
numpy oracle on a few ragged clips.

    python tools/fbank_bench.py [--check] [--iters 20] [--batch 256]
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import toolbox_for_asr_and_tts_amd as pkg  # noqa: E402
from toolbox_for_asr_and_tts_amd import _lib  # noqa: E402

SAMPLES = 480000
BYTES_PER_UTT = SAMPLES * 2 + 2998 * 80 * 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--lib", default=None, help="path of another build of libasrfe.so (A/B runs of experiment builds)")
    ap.add_argument("--bench-pcm", action="store_true", help="use bench.py's synthetic PCM (10 % leading zeros per clip)")
    args = ap.parse_args()
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    fe = pkg.WavFrontend()
    dev = "cuda"
    if args.check:
        from oracle import fbank_oracle as fo
        rng = np.random.default_rng(5)
        lens = [16000, 399, 400, 5203, 48000, 31999]
        pcm = np.zeros((len(lens), max(lens)), dtype=np.int16)
        for i, n in enumerate(lens):
            pcm[i, :n] = (rng.standard_normal(n) * 3000 + 500 * np.sin(np.arange(n) * 0.01)).astype(np.int16)
        got, nfr = fe.forward_fbank(torch.from_numpy(pcm).to(dev), lens)
        ref, _ = fo.fbank_batch(pcm, lens)
        worst = 0.0
        for i, n in enumerate(lens):
            t = fo.num_frames(n)
            assert int(nfr[i]) == t, (int(nfr[i]), t)
            if t:
                worst = max(worst, float(np.abs(got[i, :t].cpu().numpy() - ref[i][:t]).max()))
        feats, fl = fe.forward(torch.from_numpy(pcm).to(dev), lens)
        rf, rl = fo.frontend(pcm, lens)
        worst2 = 0.0
        for i in range(len(lens)):
            assert int(fl[i]) == int(rl[i])
            if rl[i]:
                worst2 = max(worst2, float(np.abs(feats[i, :rl[i]].cpu().numpy() - rf[i][:rl[i]]).max()))
        print(f"check: log-mel max err {worst:.3e}, fused feats max err {worst2:.3e}")
        assert worst < 1e-3 and worst2 < 1e-3
    fb = args.batch
    rng = np.random.default_rng(1)
    if args.bench_pcm:
        from toolbox_for_asr_and_tts_amd.synth import bench_pcm
        pcm = torch.from_numpy(bench_pcm(fb, SAMPLES, seed=4321)).to(dev)
    else:
        pcm = torch.from_numpy((rng.standard_normal((fb, SAMPLES)) * 3000).astype(np.int16)).to(dev)
    lens = torch.full((fb,), SAMPLES, dtype=torch.int32, device=dev)
    T = fe.num_frames(SAMPLES)
    Tp = fe.num_lfr_frames(T)
    logmel = torch.empty((fb, T, 80), dtype=torch.float32, device=dev)
    feats = torch.empty((fb, Tp, 560), dtype=torch.float32, device=dev)
    nfr = torch.empty((fb,), dtype=torch.int32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ctx, lib = fe.ctx, fe.ctx.lib

    def fbank():
        ctx.check(lib.asrfe_fbank(ctx.handle, pcm.data_ptr(), 0, pcm.stride(0), lens.data_ptr(), fb, T,
                                  logmel.data_ptr(), nfr.data_ptr(), stream))

    def fused():
        ctx.check(lib.asrfe_frontend(ctx.handle, pcm.data_ptr(), 0, pcm.stride(0), lens.data_ptr(), fb, Tp,
                                     feats.data_ptr(), _lib.DT_F32, nfr.data_ptr(), stream))

    # session path: a float32 session buffer, either quantised by a separate pass and fed as int16, or fed as it is
    # with the hand-off applied while the kernel loads
    from toolbox_for_asr_and_tts_amd.ingress import wire_int16
    session = (pcm.to(torch.float32) / 32767.0).contiguous()
    pcm16 = torch.empty_like(pcm)

    def wire_then_fused():
        pcm16.copy_(wire_int16(session))
        ctx.check(lib.asrfe_frontend(ctx.handle, pcm16.data_ptr(), 0, pcm16.stride(0), lens.data_ptr(), fb, Tp,
                                     feats.data_ptr(), _lib.DT_F32, nfr.data_ptr(), stream))

    def session_fused():
        ctx.check(lib.asrfe_frontend(ctx.handle, session.data_ptr(), _lib.PCM_F32_WIRE, session.stride(0),
                                     lens.data_ptr(), fb, Tp, feats.data_ptr(), _lib.DT_F32, nfr.data_ptr(), stream))

    t0 = time.time()
    while time.time() - t0 < 1.0:
        for _ in range(20):
            fbank()
        torch.cuda.synchronize()
    for _ in range(args.repeat):
        for name, fn in (("fbank", fbank), ("fused frontend", fused), ("wire + fused", wire_then_fused),
                         ("session fused", session_fused)):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / args.iters
            print(f"{name:15s} B={fb}: {us:7.1f} us  {fb * BYTES_PER_UTT / us / 1e3:7.1f} GB/s (fbank bytes)")


if __name__ == "__main__":
    main()
