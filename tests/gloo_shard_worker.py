"""Worker for test_two_rank_gloo_sharding and test_eight_rank_gloo_plans (CPU, gloo).  Mirrors bench.py's multi-rank
protocol: utterance shards, no data-path collective, barrier + MAX over ranks for time, SUM for units.

``plans`` as the first argument runs the three ways the batch is divided at the node's full width instead: weak (64
utterances per rank, every rank its own seed), strong (512 utterances split contiguously) and ragged (512 clips of
mixed length dealt out by length), each checked for being a partition and, on a sample, for giving the features the
single-process oracle gives."""
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


def plans():
    from toolbox_for_asr_and_tts_amd.engine import shard_by_length
    from toolbox_for_asr_and_tts_amd.synth import bench_pcm

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()

    def gather(t):
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return out

    # ---- weak: bench.py's default, 64 utterances per rank from seed 1234 + rank (short clips here) ----
    per_rank, n = 64, 1600
    pcm = bench_pcm(per_rank, n, seed=1234 + rank)
    mine = torch.tensor([float(np.abs(pcm.astype(np.int64)).sum()), float(fo.kaldi_fbank(pcm[rank]).sum()), per_rank],
                        dtype=torch.float64)
    got = gather(mine)
    units = torch.tensor([float(per_rank)], dtype=torch.float64)
    dist.all_reduce(units, op=dist.ReduceOp.SUM)
    if rank == 0:
        assert int(units) == per_rank * world
        assert len({float(g[0]) for g in got}) == world, "two ranks drew the same batch"
        for r in range(world):
            ref = bench_pcm(per_rank, n, seed=1234 + r)
            assert float(got[r][0]) == float(np.abs(ref.astype(np.int64)).sum())
            assert float(got[r][1]) == float(fo.kaldi_fbank(ref[r]).sum())

    # ---- strong: --global-batch 512 split contiguously ----
    lo, hi = pkg.shard_range(512, rank, world)
    spans = gather(torch.tensor([lo, hi], dtype=torch.int64))
    if rank == 0:
        edges = [(int(a), int(b)) for a, b in spans]
        assert edges[0][0] == 0 and edges[-1][1] == 512
        assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
        assert max(b - a for a, b in edges) - min(b - a for a, b in edges) <= 1

    # ---- ragged: 512 clips, U[5 s, 30 s], dealt out by length ----
    rng = np.random.default_rng(4242)
    host_lens = rng.integers(5 * 16000, 480000 + 1, size=512).tolist()
    idx = shard_by_length(host_lens, rank, world)
    padded = torch.full((512 // world + 1,), -1, dtype=torch.int64)
    padded[: len(idx)] = torch.tensor(idx, dtype=torch.int64)
    # a sample of this rank's clips through the oracle, at 1/100 of the length, keyed by the global clip index
    part = torch.zeros(512, dtype=torch.float64)
    for u in idx[:2]:
        part[u] = float(fo.kaldi_fbank(fo.synth_pcm("gauss", max(400, host_lens[u] // 100), 300 + u)).sum())
    dist.all_reduce(part, op=dist.ReduceOp.SUM)
    lists = gather(padded)
    if rank == 0:
        owned = [[int(i) for i in l if int(i) >= 0] for l in lists]
        flat = sorted(i for l in owned for i in l)
        assert flat == list(range(512)), "not a partition"
        assert max(map(len, owned)) - min(map(len, owned)) <= 1
        audio = [sum(host_lens[i] for i in l) for l in owned]
        assert max(audio) - min(audio) <= 480000, audio      # within one clip of each other
        for l in owned:
            for u in l[:2]:
                want = float(fo.kaldi_fbank(fo.synth_pcm("gauss", max(400, host_lens[u] // 100), 300 + u)).sum())
                assert float(part[u]) == want
        assert int((part != 0).sum()) == 2 * world
        print("PLANS_OK", world)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    plans() if len(sys.argv) > 1 and sys.argv[1] == "plans" else main()
