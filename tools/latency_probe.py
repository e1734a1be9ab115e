#!/usr/bin/env python3
"""Latency of one step for small batches (clips x seconds), for the plan selected by the environment
(ASRFE_UNFUSED_FFN / ASRFE_SEPARATE_QKV are read when the context is created)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import toolbox_for_asr_and_tts_amd as pkg  # noqa: E402
from toolbox_for_asr_and_tts_amd.synth import bench_pcm, encoder_state_dict  # noqa: E402


def main():
    sd = encoder_state_dict(num_blocks=50, seed=0)
    small_rows = int(sys.argv[1]) if len(sys.argv) > 1 else None   # enc_small_batch_rows (default: the library's)
    for batch, seconds in ((1, 3), (1, 10), (4, 5), (8, 10), (4, 30), (8, 30), (16, 30)):
        n = 16000 * seconds
        cfg = pkg.default_config()
        if small_rows is not None:
            cfg.enc_small_batch_rows = small_rows
        eng = pkg.AsrEngine(batch=batch, max_samples=n, state_dict=sd, cfg=cfg)
        pcm = torch.from_numpy(bench_pcm(batch, n, seed=5)).cuda()
        lens = torch.full((batch,), n, dtype=torch.int32, device="cuda")
        for _ in range(3):
            eng.run(pcm, lens)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            eng.run(pcm, lens)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        print(f"{batch:3d} x {seconds:2d} s  rows {batch * eng.tp_max:6d}: {ms:7.3f} ms", flush=True)
        del eng


if __name__ == "__main__":
    main()
