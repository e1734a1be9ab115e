#!/usr/bin/env python3
"""Times the CIF predictor + Paraformer decoder + arg-max on a batch of encoder outputs (random-init weights of the
published layout): ms per batch and the matrix rate over the GEMM FLOPs."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toolbox_for_asr_and_tts_amd as pkg  # noqa: E402
from toolbox_for_asr_and_tts_amd import _lib  # noqa: E402
from toolbox_for_asr_and_tts_amd.decoder import ParaformerHead  # noqa: E402
from toolbox_for_asr_and_tts_amd.synth import decoder_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--tp", type=int, default=500)
    ap.add_argument("--n-max", type=int, default=256)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--small-rows", type=int, default=None, help="enc_small_batch_rows of the context (0: throughput plan only)")
    args = ap.parse_args()
    cfg = _lib.default_config()
    if args.small_rows is not None:
        cfg.enc_small_batch_rows = args.small_rows
    ctx = _lib.Context(cfg)
    head = ParaformerHead(ctx)
    head.load_state_dict(decoder_state_dict(seed=1))
    g = torch.Generator(device="cpu").manual_seed(3)
    enc = torch.randn(args.batch, args.tp, 512, generator=g).cuda()
    lens = torch.full((args.batch,), args.tp, dtype=torch.int32, device="cuda")
    for _ in range(2):
        r = head.forward(enc, lens, n_max=args.n_max)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        head.forward(enc, lens, n_max=args.n_max)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    md, m = args.batch * args.n_max, args.batch * args.tp
    per_block = 2 * md * 512 * 2048 * 2 + 2 * md * 512 * 512 * 2 + 2 * m * 512 * 1024 + 4 * args.batch * 4 * args.n_max * args.tp * 128
    fl = 16 * per_block + 2 * md * 512 * 2048 * 2 + 2 * md * 512 * 8448 + 3 * 2 * m * 1536 * 512
    print(f"{args.batch} x {args.tp} frames, {args.n_max} token rows: {1e3 * dt:7.2f} ms  {fl / dt / 1e12:6.1f} TFLOP/s ({fl / 1e9:.0f} GFLOP)  "
          f"tokens {r['token_num'].float().mean().item():.1f}")


if __name__ == "__main__":
    main()
