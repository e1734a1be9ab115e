#!/usr/bin/env python3
"""Times the HiFi-GAN generator on the vocoder configuration (batch 128 of 1 s mel spectrograms, published V1 layout,
random-init weights): ms per batch, audio seconds per second, and the matrix rate over the convolutions' FLOPs."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toolbox_for_asr_and_tts_amd as pkg  # noqa: E402
from toolbox_for_asr_and_tts_amd.synth import vocoder_state_dict, vocoder_flops  # noqa: E402
from toolbox_for_asr_and_tts_amd.vocoder import HiFiGANGenerator  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--frames", type=int, default=94)   # 1 s at 24 kHz / hop 256
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--lib", default=None, help="path of another build of libasrfe.so (timing builds: ASRFE_VOC_ABLATE)")
    args = ap.parse_args()
    if args.lib:
        pkg._lib.LIB_PATH = os.path.abspath(args.lib)
    gen = HiFiGANGenerator(sample_rate=24000)
    gen.load_state_dict(vocoder_state_dict(seed=2))
    mel = torch.randn(args.batch, 80, args.frames, device="cuda")
    for _ in range(3):
        gen.forward(mel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        wav = gen.forward(mel)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    fl = vocoder_flops(args.frames) * args.batch
    print(f"batch {args.batch} x {args.frames} frames: {1e3 * dt:8.2f} ms   {args.batch * args.frames * 256 / 24000 / dt:9.1f} s of audio per s   "
          f"{fl / dt / 1e12:7.1f} TFLOP/s ({fl / 1e9:.0f} GFLOP)   finite {bool(torch.isfinite(wav).all())}")


if __name__ == "__main__":
    main()
