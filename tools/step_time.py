#!/usr/bin/env python3
"""Times the engine step (64 x 30 s, 50 blocks, two streams) with a chosen build of the library: same-box A/B.

    python tools/step_time.py [--lib PATH] [--iters 40] [--streams 2]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toolbox_for_asr_and_tts_amd as pkg  # noqa: E402
from toolbox_for_asr_and_tts_amd.synth import bench_pcm, encoder_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--iters", type=int, default=40)
ap.add_argument("--streams", type=int, default=2)
args = ap.parse_args()
if args.lib:
    pkg._lib.LIB_PATH = os.path.abspath(args.lib)
pkg.load()
sd = encoder_state_dict(num_blocks=50, seed=0)
pcm = torch.from_numpy(bench_pcm(64, 480000, seed=1234)).cuda()
lens = torch.full((64,), 480000, dtype=torch.int32, device="cuda")
eng = pkg.AsrEngine(batch=64, max_samples=480000, state_dict=sd, n_streams=args.streams)
for _ in range(5):
    eng.run(pcm, lens)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        eng.run(pcm, lens)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / args.iters)
print(f"lib {args.lib or 'in-tree'} streams {args.streams}: {best:.3f} ms/step  {64e3 / best:.0f} utt/s")
