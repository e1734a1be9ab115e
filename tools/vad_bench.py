#!/usr/bin/env python3
"""Times the FSMN voice-activity network (asrfe_vad_forward, random-init weights of the published layout): a batch of
streams of 30 s each (throughput) and one 400 ms chunk with its cache (what a streaming session asks per chunk), plus the
whole per-chunk generate() call of the model object (frontend + network + detector)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toolbox_for_asr_and_tts_amd as pkg  # noqa: E402
from toolbox_for_asr_and_tts_amd import _lib  # noqa: E402
from toolbox_for_asr_and_tts_amd.synth import bench_pcm, vad_state_dict  # noqa: E402


def timed(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--frames", type=int, default=2998)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    cfg = _lib.default_config()
    cfg.lfr_m, cfg.lfr_n, cfg.enc_input_size = 5, 1, 400
    ctx = _lib.Context(cfg)
    net = pkg.FsmnVadNet(ctx).load_state_dict(vad_state_dict(seed=1))
    g = torch.Generator(device="cpu").manual_seed(3)
    flop_row = 2 * (400 * 140 + 140 * 250 + 4 * (250 * 128 + 128 * 20 + 128 * 250) + 250 * 140 + 140 * 248)
    x = torch.randn(args.batch, args.frames, 400, generator=g).cuda()
    cache = net.new_cache(args.batch, "cuda")
    dt = timed(lambda: net(x, cache=cache), args.iters)
    rows = args.batch * args.frames
    print(f"{args.batch} streams x {args.frames} frames: {1e3 * dt:7.3f} ms  {rows / dt / 1e6:7.2f} M frames/s  "
          f"{rows * flop_row / dt / 1e12:6.2f} TFLOP/s (model FLOPs)  = {rows * 0.01 / dt:9.0f} x real time")
    x1 = torch.randn(1, 40, 400, generator=g).cuda()
    c1 = net.new_cache(1, "cuda")
    dt1 = timed(lambda: net(x1, cache=c1), 50)
    print(f"one 400 ms chunk (40 frames) with its cache: {1e6 * dt1:7.1f} us")
    model = pkg.FsmnVadModel({}, (np.zeros(400, np.float32), np.ones(400, np.float32)), vad_state_dict(seed=1))
    audio = bench_pcm(1, 16000 * 20, 5)[0].astype(np.float32) / 32768.0
    st = {}
    chunks = [audio[i:i + 6400] for i in range(0, len(audio), 6400)]
    model.generate(input=chunks[0], cache=st, is_final=False, chunk_size=400)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for c in chunks[1:]:
        model.generate(input=c, cache=st, is_final=False, chunk_size=400)
    torch.cuda.synchronize()
    print(f"generate() per 400 ms chunk (fbank + level + LFR/CMVN + network + detector, host round trips included): "
          f"{1e3 * (time.perf_counter() - t0) / (len(chunks) - 1):6.2f} ms")


if __name__ == "__main__":
    main()
