"""Worker for test_two_rank_gloo_sharding (CPU, gloo).  Mirrors bench.py's multi-rank protocol: contiguous
utterance shards, no data-path collective, barrier + MAX over ranks for time, SUM for units."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import fbank_oracle as fo  # noqa: E402
import toolbox_for_asr_and_tts_amd as pkg  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n_utts = 6
    lens = [4000, 1600, 3200, 400, 2400, 1000]
    lo, hi = pkg.shard_range(n_utts, rank, world)
    mine = torch.zeros(n_utts, 2, dtype=torch.float64)
    for u in range(lo, hi):
        m = fo.kaldi_fbank(fo.synth_pcm("gauss", lens[u], 100 + u))
        mine[u, 0] = m.shape[0]
        mine[u, 1] = float(m.sum())
    dist.barrier()
    dist.all_reduce(mine, op=dist.ReduceOp.SUM)  # test-side gather only; the product path exchanges nothing
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    units = torch.tensor([float(hi - lo)], dtype=torch.float64)
    dist.all_reduce(units, op=dist.ReduceOp.SUM)
    if rank == 0:
        ref = torch.zeros(n_utts, 2, dtype=torch.float64)
        for u in range(n_utts):
            m = fo.kaldi_fbank(fo.synth_pcm("gauss", lens[u], 100 + u))
            ref[u, 0], ref[u, 1] = m.shape[0], float(m.sum())
        assert torch.equal(mine, ref), (mine, ref)
        assert abs(float(t) - 0.1 * world) < 1e-12 and int(units) == n_utts
        print("SHARD_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
