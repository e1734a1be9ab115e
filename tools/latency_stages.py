#!/usr/bin/env python3
"""Per-kernel-kind time of one small-batch step (HIP events between launches), for the plan selected by the
environment (ASRFE_UNFUSED_FFN / ASRFE_SEPARATE_QKV): where a single short clip's latency goes."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import toolbox_for_asr_and_tts_amd as pkg  # noqa: E402
from toolbox_for_asr_and_tts_amd.synth import bench_pcm, encoder_state_dict  # noqa: E402


def main():
    sd = encoder_state_dict(num_blocks=50, seed=0)
    for batch, seconds in ((1, 3), (4, 5)):
        n = 16000 * seconds
        eng = pkg.AsrEngine(batch=batch, max_samples=n, state_dict=sd)
        pcm = torch.from_numpy(bench_pcm(batch, n, seed=5)).cuda()
        lens = torch.full((batch,), n, dtype=torch.int32, device="cuda")
        for _ in range(3):
            eng.run(pcm, lens)
        torch.cuda.synchronize()
        eng.ctx.profile_begin(4096)
        for _ in range(5):
            eng.run(pcm, lens)
        prof = eng.ctx.profile_end()
        print(f"{batch} x {seconds} s, rows {batch * eng.tp_max}:",
              {k: f"{v[0] // 5} x {1e3 * v[1] / v[0]:.1f} us" for k, v in prof.items()}, flush=True)
        del eng


if __name__ == "__main__":
    main()
