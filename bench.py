#!/usr/bin/env python3
"""Benchmark of the hot path: 16 kHz PCM -> fbank/LFR/CMVN -> Paraformer-large SAN-M encoder (50 blocks).

    python bench.py --gpus N --steps K --warmup W

A step is one pass of frontend + encoder over one batch of 64 synthetic 30 s utterances per GPU (the
per-GPU slice of BASELINE.json's "batch 512 sharded 64/GPU" configuration; inputs are resident in HBM
before the timed region).  Utterances are independent, so ranks exchange nothing on the data path: only a
barrier and a MAX over ranks of the elapsed time (RCCL).  N > 1 runs either under a launcher that sets
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (python -m torch.distributed.run ...), or plainly as
``python bench.py --gpus N``: the process then starts N children of itself, one per GPU, before it touches
the GPU, relays rank 0's JSON line and fails if a child fails or fewer than N devices are visible.
``--scaling strong`` keeps the global batch fixed (--global-batch, default 512) and splits it over the ranks.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      : the dominant kernel (post_attn_kernel: everything of a layer behind attention, fp16 MFMA
                  operands) priced against the dense 16-bit MFMA peak from HIP-event timings taken inside
                  this process,
  roofline_fbank: the fbank kernel on 256 x 30 s (BASELINE.json configs[1]) priced against HBM peak,
  cpu_baseline  : the CPU oracle (a port: the reference's own CPU stack is not installable offline)
                  timed on a bounded sample on this host's cores,
  stages        : per-kernel-kind time per step from the same event pass.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SAMPLES = 480000          # 30 s @ 16 kHz
T_FRAMES = 2998
TP = 500
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16, MI355X
HBM_PEAK_GBPS = 8000.0
FBANK_BYTES_PER_UTT = 960000 + 959360   # int16 PCM in + fp32 log-mel out
GEMM_FLOP_PER_UTT = 157.4e9             # QKV + out-proj + FFN over 50 layers at T' = 500
ALL_FLOP_PER_UTT = 183.2e9              # + attention matmuls + FSMN


_COUNTERS = None


def counter_profile():
    """The newest committed counter pass (profiles/r*_hbm_traffic.json; the counters need rocprofv3 runs of their own,
    so they cannot be taken inside this process) IF it was taken with the kernel sources this tree holds: the file
    records a hash of csrc/ (tools/buildtag.py) and a file of another build is not quoted.  Returns (data or None,
    source record for the JSON line)."""
    global _COUNTERS
    if _COUNTERS is None:
        import glob
        from tools.buildtag import csrc_tag
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))
        src = {"file": None, "build": None, "csrc_tag": None, "this_build_csrc_tag": csrc_tag(), "used": False}
        data = None
        if files:
            d = json.load(open(files[-1]))
            src.update(file=os.path.relpath(files[-1], ROOT), build=d.get("build"), csrc_tag=d.get("csrc_tag"))
            if d.get("csrc_tag") == src["this_build_csrc_tag"]:
                data, src["used"] = d, True
            else:
                src["note"] = "counter file belongs to another build of the kernels: not quoted"
        _COUNTERS = (data, src)
    return _COUNTERS


def hbm_traffic(kernel):
    """HBM bytes per launch of a kernel (FETCH_SIZE corrected + WRITE_SIZE) from the committed counter pass of this
    build; None if there is none."""
    d, _ = counter_profile()
    k = None if d is None else d.get("kernels", {}).get(kernel)
    return None if k is None else k["fetch_bytes"] + k["write_bytes"]


def side_traffic(program):
    """HBM bytes per pass of a whole side program ("decoder", "vocoder": tools/profile_side.sh) from the same file."""
    d, _ = counter_profile()
    from tools.buildtag import side_tag
    k = None if d is None or d.get("side_tag") != side_tag() else d.get("side", {}).get(program)
    return None if k is None else k["fetch_bytes"] + k["write_bytes"]


def mfma_pipe_busy(kernel):
    """Matrix-pipe busy cycles over kernel cycles (SQ_VALU_MFMA_BUSY_CYCLES per SIMD / GRBM_GUI_ACTIVE per XCD) from the
    same file; a profile-time figure, not measured in this run.  None if there is none."""
    d, _ = counter_profile()
    k = None if d is None else d.get("mfma_pipe", {}).get(kernel)
    return None if k is None else k["busy_frac_of_kernel_cycles"]


def fbank_units():
    """LDS / vector-ALU busy fractions of the fbank kernel from the newest committed counter pass of this build
    (profiles/r*_fbank_units.json, tools/make_fbank_units_json.py); None if there is none."""
    import glob
    from tools.buildtag import csrc_tag
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fbank_units.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))
    if d.get("csrc_tag") != csrc_tag():
        return None
    d["file"] = os.path.relpath(files[-1], ROOT)
    return d


def cpu_model():
    """Model name of the first CPU with its family / model / stepping numbers and clock (a virtualised host often
    reports a bare vendor name: the numbers still identify the part)."""
    want = {"model name": None, "cpu family": None, "model": None, "stepping": None, "cpu MHz": None}
    try:
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k = k.strip()
            if k in want and want[k] is None:
                want[k] = v.strip()
            if all(x is not None for x in want.values()):
                break
    except OSError:
        pass
    if want["model name"] is None:
        return "unknown"
    extra = ", ".join(f"{k.replace('cpu ', '')} {want[k]}" for k in ("cpu family", "model", "stepping", "cpu MHz") if want[k])
    return want["model name"] + (f" ({extra})" if extra else "")


def gemm_flops(m, n, k):
    return 2.0 * m * n * k


def cpu_baseline(sd, n_clips: int, batch: int = 8):
    """Times the CPU oracle on n_clips 30 s utterances, `batch` at a time (frontend in numpy float64, encoder in
    torch fp32 on the torch thread pool): a bounded sample of the bench workload, about 10-20 s of CPU work."""
    from oracle import fbank_oracle as fo
    from oracle import sanm_oracle as so
    from toolbox_for_asr_and_tts_amd.synth import bench_pcm

    cfg = so.SanmConfig(num_blocks=50)
    t_front = t_enc = 0.0
    done = 0
    while done < n_clips:
        b = min(batch, n_clips - done)
        pcm = bench_pcm(b, SAMPLES, seed=99 + done)
        t0 = time.perf_counter()
        feats, lens = fo.frontend(pcm, [SAMPLES] * b)
        t1 = time.perf_counter()
        with torch.no_grad():
            so.encoder_forward(torch.from_numpy(feats).float(), torch.from_numpy(lens), sd, cfg)
        t2 = time.perf_counter()
        t_front += t1 - t0
        t_enc += t2 - t1
        done += b
    # one thread: a single clip through the same chain
    nthr = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        pcm1 = bench_pcm(1, SAMPLES, seed=7)
        t0 = time.perf_counter()
        f1, l1 = fo.frontend(pcm1, [SAMPLES])
        with torch.no_grad():
            so.encoder_forward(torch.from_numpy(f1).float(), torch.from_numpy(l1), sd, cfg)
        t_one = time.perf_counter() - t0
    finally:
        torch.set_num_threads(nthr)
    cxx = None
    try:
        cxx = cxx_cpu_baseline(sd, nthr)
    except Exception as e:  # (the C++ restatement is a second opinion: its absence must not cost the bench line)
        cxx = {"error": f"{type(e).__name__}: {e}"}
    finally:
        torch.set_num_threads(nthr)  # (the restatement's one-thread leg sets the OpenMP thread count torch shares)
    return {
        "cxx_port": cxx,
        "value": n_clips / (t_front + t_enc),
        "one_thread_value": 1.0 / t_one,
        "unit": "utt/s",
        "cores": int(nthr),
        "host_cpus": os.cpu_count(),
        "cpu_model": cpu_model(),
        "kind": "port",
        "sample": f"{n_clips} x 30 s utterances ({batch} per call) through oracle/fbank_oracle.py (numpy f64, 1 thread) "
                  f"and oracle/sanm_oracle.py (torch fp32, {nthr} threads), 50 blocks",
        "frontend_s": t_front,
        "encoder_s": t_enc,
    }


def cxx_cpu_baseline(sd, threads: int):
    """The same chain through the C++ restatement of the library's own entry points (oracle/cpu_abi: asrfe_frontend +
    asrfe_encoder_forward on host pointers, OpenMP, AVX2): 4 clips of 30 s on `threads` threads, then one 10 s clip on
    one thread scaled to 30 s by its frame count (the encoder's cost per frame grows with the clip, so this flatters
    the single thread slightly)."""
    from tests.cpu_abi_util import CpuModel
    from toolbox_for_asr_and_tts_amd.synth import bench_pcm
    m = CpuModel(num_blocks=50, state_dict=sd, threads=threads)
    pcm = bench_pcm(4, SAMPLES, seed=199)
    t0 = time.perf_counter()
    feats, lens = m.frontend(pcm, [SAMPLES] * 4)
    t1 = time.perf_counter()
    m.encoder(feats, lens)
    t2 = time.perf_counter()
    m.set_threads(1)
    short = SAMPLES // 3
    t3 = time.perf_counter()
    f1, l1 = m.frontend(pcm[:1, :short], [short])
    m.encoder(f1, l1)
    t_one = (time.perf_counter() - t3) * TP / float(l1[0])
    m.close()
    return {"value": 4.0 / (t2 - t0), "one_thread_value": 1.0 / t_one, "unit": "utt/s", "cores": threads, "kind": "port",
            "frontend_s": t1 - t0, "encoder_s": t2 - t1,
            "sample": f"4 x 30 s utterances in one call on {threads} threads; one 10 s clip on 1 thread scaled to 30 s"}


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int) -> int:
    """``python bench.py --gpus N`` without a launcher: start N copies of this script, one rank per GPU, relay rank 0's
    output and return non-zero if any rank fails.  Runs BEFORE this process makes any GPU call (counting devices does
    not initialise the GPU) and starts fresh child processes: a process that has touched the GPU is never replaced."""
    import subprocess

    rehearsal = os.environ.get("ASRFE_BENCH_REHEARSAL") == "1"  # every rank on cuda:0 of a one-GPU box (gloo)
    n_dev = torch.cuda.device_count()
    if n_dev < n and not rehearsal:
        print(f"bench.py: --gpus {n} but only {n_dev} GPU(s) are visible", file=sys.stderr)
        return 2
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for r, p in enumerate(procs):
        code = p.wait()
        if code != 0:
            print(f"bench.py: rank {r} exited with status {code}", file=sys.stderr)
            rc = rc or (code if 0 < code < 256 else 1)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU per step")
    ap.add_argument("--fbank-batch", type=int, default=256)
    ap.add_argument("--cpu-clips", type=int, default=48)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stages", action="store_true")
    ap.add_argument("--streams", type=int, default=2, help="HIP streams (batch chunks) per GPU")
    ap.add_argument("--graph", type=int, default=1, help="1: replay the step from a captured HIP graph (falls back to plain "
                    "launches if the capture fails), 0: plain launches")
    ap.add_argument("--buckets", type=int, default=4, help="length buckets of the ragged side run (0: skip it)")
    ap.add_argument("--no-extras", action="store_true", help="skip the ragged and host-copy side runs")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --batch utterances per GPU; strong: --global-batch utterances split over the GPUs")
    ap.add_argument("--global-batch", type=int, default=512, help="utterances per step over all GPUs (strong scaling)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus))  # this process never touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: a measurement of {world} rank(s) would be "
                         f"reported as one of {args.gpus}")
    if args.scaling == "strong":
        from toolbox_for_asr_and_tts_amd.engine import shard_range
        lo, hi = shard_range(args.global_batch, rank, world)
        args.batch = hi - lo
        if args.batch < 1:
            raise SystemExit("--global-batch must give every rank at least one utterance")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # ASRFE_BENCH_REHEARSAL=1: a dry run of the multi-rank path on a box with ONE GPU (every rank on cuda:0, gloo
    # instead of RCCL, which refuses two ranks on one device); its numbers mean nothing, it only exercises the code.
    rehearsal = world > 1 and os.environ.get("ASRFE_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    backend = None
    # ASRFE_BENCH_FORCE_DIST=1: a single rank still forms an RCCL process group and goes through the barrier and the
    # reductions (the multi-rank code path of a real node, exercised on a box with one GPU)
    force_dist = world == 1 and os.environ.get("ASRFE_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist  # RCCL; used for the barrier and the MAX over ranks only

        if force_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        backend = "gloo" if rehearsal else "nccl"
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import toolbox_for_asr_and_tts_amd as pkg
    from toolbox_for_asr_and_tts_amd.synth import bench_pcm, encoder_state_dict

    sd = encoder_state_dict(num_blocks=50, seed=0)
    eng = pkg.AsrEngine(batch=args.batch, max_samples=SAMPLES, state_dict=sd, n_streams=args.streams)
    pcm = torch.from_numpy(bench_pcm(args.batch, SAMPLES, seed=1234 + rank)).to(dev)
    lens = torch.full((args.batch,), SAMPLES, dtype=torch.int32, device=dev)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    step = lambda: eng.run(pcm, lens)
    graph_replay = False
    plain_out = None  # the output of one plain-launch step, to hold the replayed graph's output against
    if args.graph:
        # the step's ~105 launches on its two streams as one HIP graph: the same kernels on the same buffers, launched
        # by one call (the launch-bound part of a 12 ms step is about 1 %)
        eng.run(pcm, lens)
        torch.cuda.synchronize()
        plain_out = eng.out.clone()
        try:
            graph = torch.cuda.CUDAGraph()
            # (thread-local capture: a process group's watchdog thread may query its events while this thread captures)
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                eng.run(pcm, lens)
            graph.replay()
            torch.cuda.synchronize()
            step = graph.replay
            graph_replay = True
        except Exception as e:  # (a box whose runtime refuses the capture still gets its measurement)
            print(f"bench.py: HIP graph capture failed ({type(e).__name__}: {e}); timing plain launches", file=sys.stderr)
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ranks_seen = 1
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ones = torch.ones(1, dtype=torch.int32, device="cpu" if rehearsal else dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)  # every rank that took part in the timed region counts itself
        ranks_seen = int(ones.item())
        counts = torch.tensor([args.batch], dtype=torch.int64, device="cpu" if rehearsal else dev)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        global_batch = int(counts.item())
        dist.barrier()
    else:
        global_batch = args.batch
    out_ok = bool(torch.isfinite(eng.out).all().item())
    # what the last timed step left in the output buffer is, bit for bit, what a step of plain launches computes
    replay_equals_plain = bool(torch.equal(eng.out, plain_out)) if (graph_replay and plain_out is not None) else None
    del plain_out

    result = None
    if rank == 0:
        utts = global_batch * args.steps
        value = utts / elapsed
        result = {
            "metric": "utterances/sec (30 s @16 kHz) fbank+Paraformer encoder",
            "value": value,
            "unit": "utt/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "collective_backend": backend,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {
                "workload": f"fbank+LFR+CMVN+SAN-M encoder (50 blocks, d=512), {args.batch} x 30 s int16 utterances "
                            f"per GPU per step (BASELINE configs[3] slice: 64/GPU), random-init N(0,0.02) weights",
                "batch_per_gpu": args.batch if args.scaling == "weak" else f"{global_batch} / {world}",
                "global_batch": global_batch,
                "parallelism": f"utterance-sharded x{world}, no collective",
                "streams_per_gpu": args.streams,
                "graph_replay": graph_replay,
            },
            "output_finite": out_ok,
            "graph_replay_output_equals_plain_launches": replay_equals_plain,
            # what one rank has of the host: the step is launched from one thread, but a rank that shares a core with
            # another shows up as launch gaps at N = 8
            "host": {"cores_per_rank": len(os.sched_getaffinity(0)), "cpus_on_box": os.cpu_count(),
                     "torch_threads_per_rank": torch.get_num_threads()},
            "achieved_tflops_all": value * ALL_FLOP_PER_UTT / 1e12 / world,
            "counter_profile": counter_profile()[1],
        }
        if ranks_seen != world:
            raise SystemExit(f"{ranks_seen} rank(s) answered the all-reduce, {world} expected")

    if rank == 0 and world == 1 and not args.no_extras:
        # ---- side runs, reported beside the headline number and never part of it --------------------------
        def timed(fn, n):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n

        # (1) the same step with the PCM starting in pinned host memory: one 16-bit copy over PCIe per step, first
        #     in line with the step, then staged one batch ahead on a copy stream (AsrEngine.stage / run_staged)
        pcm_host = pcm.cpu().pin_memory()
        lens_host = lens.cpu().pin_memory()
        pcm_in = torch.empty_like(pcm)

        def step_h2d():
            pcm_in.copy_(pcm_host, non_blocking=True)
            eng.run(pcm_in, lens)

        t_h2d = timed(step_h2d, args.steps)
        slot = [0]
        eng.stage(pcm_host, lens_host, 0)

        def step_staged():
            eng.stage(pcm_host, lens_host, slot[0] ^ 1)
            eng.run_staged(slot[0])
            slot[0] ^= 1

        t_staged = timed(step_staged, args.steps)
        result["host_copy_inclusive"] = {
            "in_line": {"utt_per_s": args.batch / t_h2d, "ms_per_step": 1e3 * t_h2d},
            "staged_one_batch_ahead": {"utt_per_s": args.batch / t_staged, "ms_per_step": 1e3 * t_staged},
            "note": "int16 PCM copied from pinned host memory every step",
        }
        del pcm_host, pcm_in
        # (2) clips of mixed length, U[5 s, 30 s]: padded to the longest clip against length-bucketed.  A bucket
        #     must still fill the GPU (one 128-row workgroup per CU), so this run uses 4 x the headline batch.
        if args.buckets > 0:
            rb = 4 * args.batch
            eng_r = pkg.AsrEngine(batch=rb, max_samples=SAMPLES, state_dict=sd, n_streams=args.streams,
                                  n_buckets=args.buckets)
            pcm_r = torch.from_numpy(bench_pcm(rb, SAMPLES, seed=77)).to(dev)
            rng = np.random.default_rng(4242)
            host_lens = rng.integers(5 * 16000, SAMPLES + 1, size=rb).tolist()
            lens_r = torch.tensor(host_lens, dtype=torch.int32, device=dev)
            n = max(2, args.steps // 2)
            t_pad = timed(lambda: eng_r.run(pcm_r, lens_r), n)
            t_bkt = timed(lambda: eng_r.run_ragged(pcm_r, lens_r, host_lens), n)
            audio_s = sum(host_lens) / 16000.0
            result["ragged"] = {
                "clips": rb, "lens": "U[5 s, 30 s], numpy default_rng(4242)", "mean_clip_s": audio_s / rb,
                "padded_utt_per_s": rb / t_pad, "bucketed_utt_per_s": rb / t_bkt,
                "padded_audio_s_per_s": audio_s / t_pad, "bucketed_audio_s_per_s": audio_s / t_bkt,
                "buckets": args.buckets,
            }
            del eng_r, pcm_r

        # (2b) the same step with bf16 MFMA operands (the operand type BASELINE.json's encoder configuration names; the
        #      default here is fp16 because bf16 rounding alone exceeds the 1e-2 parity bar): same kernels, same rate
        from toolbox_for_asr_and_tts_amd import _lib as _l
        cfg_b = pkg.default_config()
        cfg_b.enc_compute_dtype = _l.DT_BF16
        eng_b = pkg.AsrEngine(batch=args.batch, max_samples=SAMPLES, cfg=cfg_b, state_dict=sd, n_streams=args.streams)
        t_b = timed(lambda: eng_b.run(pcm, lens), args.steps)
        result["dtype_alt"] = {"bf16": {"utt_per_s": args.batch / t_b, "ms_per_step": 1e3 * t_b,
                                        "note": "enc_compute_dtype = ASRFE_DT_BF16, same step; parity 6e-2 instead of 1e-2"}}
        del eng_b

        # (2c) the step behind the encoder (CIF predictor + 16-block decoder + arg-max, up to 256 tokens per clip) on the
        #      encoder output of the headline step: its own time, and the rate of the whole chain PCM -> token ids
        from toolbox_for_asr_and_tts_amd.decoder import ParaformerHead
        from toolbox_for_asr_and_tts_amd.synth import decoder_state_dict
        head = ParaformerHead(eng.ctx)
        head.load_state_dict(decoder_state_dict(seed=1))
        eng.run(pcm, lens)
        r = head.forward(eng.out, eng.feat_lens, n_max=256)
        torch.cuda.synchronize()
        tok = r["token_num"].cpu().tolist()
        t_head = timed(lambda: head.forward(eng.out, eng.feat_lens, n_max=256), max(2, args.steps // 2))

        def step_text():
            o, l = eng.run(pcm, lens)
            head.forward(o, l, n_max=256)

        t_text = timed(step_text, max(2, args.steps // 2))
        # matrix work of the step behind the encoder: per decoder block the feed-forward (512 -> 2048 -> 512), the query
        # and output projections on the token rows, the key / value projection on the encoder frames and the cross
        # attention; the closing feed-forward, the vocabulary projection and the predictor's three products
        md, mf = args.batch * 256, args.batch * TP
        dec_flop = (16 * (2 * gemm_flops(md, 2048, 512) + 2 * gemm_flops(md, 512, 512) + gemm_flops(mf, 1024, 512)
                          + args.batch * 4 * 2 * gemm_flops(256, TP, 128))
                    + 2 * gemm_flops(md, 2048, 512) + gemm_flops(md, 8448, 512) + 3 * gemm_flops(mf, 512, 1536))
        result["roofline_decoder"] = {
            "kernel": "CIF predictor + 16 decoder blocks + vocabulary projection with arg-max epilogue, 64 x 256 token rows "
                      "over 64 x 500 encoder frames (about 165 launches; the products run on gemm_bf16_kernel)",
            "bound": "mfma", "achieved": dec_flop / t_head / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": dec_flop / t_head / 1e12 / MFMA_BF16_PEAK_TFLOPS, "traffic": side_traffic("decoder"),
            "flop_per_step": dec_flop,
            "note": "all products of one step over the step's wall time (memory-bound passes included in the time)",
        }
        result["with_decoder"] = {
            "head_ms_per_step": 1e3 * t_head, "chain_ms_per_step": 1e3 * t_text, "chain_utt_per_s": args.batch / t_text,
            "tokens_per_clip_mean": sum(tok) / len(tok), "n_max": 256,
            "note": "CIF predictor + Paraformer decoder (16 blocks, vocabulary 8404, random-init weights) + arg-max behind the "
                    "headline step; never part of `value`",
        }
        del head

        # (2d) the vocoder configuration (BASELINE.json configs[4]): HiFi-GAN generator, published V1 layout (80 mel ->
        #      hop 256, i.e. 24 kHz at 93.75 frames per second), batch 128 of 1 s spectrograms, random-init weights
        from toolbox_for_asr_and_tts_amd.synth import vocoder_flops, vocoder_state_dict
        from toolbox_for_asr_and_tts_amd.vocoder import HiFiGANGenerator
        voc = HiFiGANGenerator(ctx=eng.ctx, sample_rate=24000)
        voc.load_state_dict(vocoder_state_dict(seed=2))
        mel = torch.randn(128, 80, 94, device=dev)
        t_voc = timed(lambda: voc.forward(mel), 5)
        vfl = 128 * vocoder_flops(94)
        result["vocoder"] = {
            "workload": "HiFi-GAN generator (512 channels, rates 8-8-2-2, residual kernels 3-7-11), 128 x 94 mel frames -> 128 x 24064 samples",
            "ms_per_batch": 1e3 * t_voc, "audio_s_per_s": 128 * 94 * 256 / 24000.0 / t_voc,
            "roofline": {"bound": "mfma", "achieved": vfl / t_voc / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": vfl / t_voc / 1e12 / MFMA_BF16_PEAK_TFLOPS, "traffic": side_traffic("vocoder"),
                         "note": "all convolutions of one batch (implicit-GEMM FLOPs) over the whole forward's wall time"},
        }
        del voc, mel

        # (3) latency of ONE 3 s clip (the shape of the reference's own CPU-runnable case, BASELINE configs[0]): eager
        #     launches and the same step replayed from a captured HIP graph
        eng_1 = pkg.AsrEngine(batch=1, max_samples=48000, state_dict=sd)
        pcm_1 = torch.from_numpy(bench_pcm(1, 48000, seed=99)).to(dev)
        lens_1 = torch.full((1,), 48000, dtype=torch.int32, device=dev)
        t_eager = timed(lambda: eng_1.run(pcm_1, lens_1), 20)
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            eng_1.run(pcm_1, lens_1)
        t_graph = timed(g1.replay, 20)
        head_1 = ParaformerHead(eng_1.ctx)
        head_1.load_state_dict(decoder_state_dict(seed=1))

        def clip_to_ids():
            o, l = eng_1.run(pcm_1, lens_1)
            head_1.forward(o, l)

        t_ids = timed(clip_to_ids, 20)
        result["single_clip_3s"] = {"eager_ms": 1e3 * t_eager, "graph_ms": 1e3 * t_graph, "to_token_ids_ms": 1e3 * t_ids,
                                    "note": "one 3 s utterance per step (T' = 50 rows): latency, not throughput; "
                                            "to_token_ids_ms adds the CIF predictor + 16-block decoder + arg-max"}
        del eng_1, g1, head_1

    if rank == 0 and not args.no_stages:
        # ---- per-kernel HIP-event pass (same process, same inputs, after the timed region) ----------
        torch.cuda.synchronize()
        eng.run_single_stream(pcm, lens)  # allocates its workspace on first use
        torch.cuda.synchronize()
        eng.ctx.profile_begin(4096)
        nprof = 3
        for _ in range(nprof):
            eng.run_single_stream(pcm, lens)
        prof = eng.ctx.profile_end()
        stages = {k: {"launches_per_step": v[0] // nprof, "ms_per_step": v[1] / nprof} for k, v in prof.items()}
        m = args.batch * TP
        # QKV projections that run as launches of their own (layer 0 always does); the rest run inside the
        # post-attention kernel of the layer before
        n_qkv = stages.get("gemm_qkv", {}).get("launches_per_step", 0)
        flops = {
            "gemm_qkv": max(n_qkv - 1, 0) * gemm_flops(m, 1536, 512) + gemm_flops(m, 1536, 560),
            "gemm_out": 50 * gemm_flops(m, 512, 512),
            "gemm_ffn1": 50 * gemm_flops(m, 2048, 512),
            "gemm_ffn2": 50 * gemm_flops(m, 512, 2048),
            "post_attn": 50 * (gemm_flops(m, 512, 512) + gemm_flops(m, 2048, 512) + gemm_flops(m, 512, 2048))
                         + (50 - n_qkv) * gemm_flops(m, 1536, 512),
            "attention": 50 * args.batch * 4 * 2 * gemm_flops(TP, TP, 128),
        }
        for k, f in flops.items():
            if k in stages:
                stages[k]["tflops"] = f / (stages[k]["ms_per_step"] * 1e-3) / 1e12
        result["stages"] = stages
        gemm_kinds = [k for k in ("gemm_qkv", "gemm_out", "gemm_ffn1", "gemm_ffn2", "post_attn") if k in stages]
        gemm_ms = sum(stages[k]["ms_per_step"] for k in gemm_kinds)
        gemm_fl = sum(flops[k] for k in gemm_kinds)
        result["gemm_mfma_util"] = gemm_fl / (gemm_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS
        # the dominant kernel: everything after attention in one launch (Wo + FSMN + residual, norm2, W1, ReLU, W2, next
        # norm1 and the next layer's QKV projection; the last layer's launch has no projection, so flop_per_launch is
        # the mean over the 50 launches of a step, as launch_us is)
        dom = "post_attn" if "post_attn" in stages else "gemm_ffn1"
        if dom in stages:
            st = stages[dom]
            fl = flops[dom] / 50
            ach = fl / (st["ms_per_step"] / st["launches_per_step"] * 1e-3) / 1e12
            result["roofline"] = {
                "kernel": ("post_attn16_kernel<f16> (v_mfma_f32_16x16x32; 32000 rows per launch: att Wo^T + FSMN + residual, LayerNorm, "
                           "W1 [2048 x 512], bias + ReLU, W2 [512 x 2048], bias + residual, next LayerNorm, "
                           "next layer's Wqkv [1536 x 512] + bias in 49 of 50 launches)") if dom == "post_attn" else
                          "gemm_bf16_kernel<EPI_BF16_RELU> (FFN1: [32000 x 512] x [512 x 2048], bias + ReLU, 16-bit out)",
                "bound": "mfma",
                "achieved": ach,
                "peak": MFMA_BF16_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": ach / MFMA_BF16_PEAK_TFLOPS,
                "traffic": hbm_traffic("post_attn") if dom == "post_attn" else None,
                "mfma_pipe_busy_profiled": mfma_pipe_busy("post_attn") if dom == "post_attn" else None,
                "launch_us": 1e3 * st["ms_per_step"] / st["launches_per_step"],
                "flop_per_launch": fl,
            }

        if "attention" in stages:
            st = stages["attention"]
            fl = flops["attention"] / 50
            us = 1e3 * st["ms_per_step"] / st["launches_per_step"]
            tr = hbm_traffic("attention")
            result["roofline_attention"] = {
                "kernel": "attn_kernel<f16> (64 x 4 heads x 500 x 500 x 128: Q K^T, online softmax, P V, then the FSMN "
                          "memory block of the same rows)",
                "bound": "mfma",
                "achieved": fl / (us * 1e-6) / 1e12,
                "peak": MFMA_BF16_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": fl / (us * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                "traffic": tr,
                "hbm_gbps_from_traffic": None if tr is None else tr / (us * 1e-6) / 1e9,
                "mfma_pipe_busy_profiled": mfma_pipe_busy("attention"),
                "launch_us": us,
                "flop_per_launch": fl,
            }

        # ---- fbank-only stage: 256 x 30 s -> 80-dim log-mel (HBM roofline) ---------------------------
        fb = args.fbank_batch
        del pcm
        pcm_f = torch.from_numpy(bench_pcm(fb, SAMPLES, seed=4321)).to(dev)
        lens_f = torch.full((fb,), SAMPLES, dtype=torch.int32, device=dev)
        logmel = torch.empty((fb, T_FRAMES, 80), dtype=torch.float32, device=dev)
        nfr = torch.empty((fb,), dtype=torch.int32, device=dev)
        import ctypes as C

        def fbank_once():
            eng.ctx.check(eng.ctx.lib.asrfe_fbank(eng.ctx.handle, pcm_f.data_ptr(), 0, pcm_f.stride(0), lens_f.data_ptr(),
                                                  fb, T_FRAMES, logmel.data_ptr(), nfr.data_ptr(),
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream)))

        # steady state of THIS kernel: the clocks need a moment to settle after the encoder steps before
        t_warm = time.perf_counter()
        while time.perf_counter() - t_warm < 0.5:
            for _ in range(20):
                fbank_once()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            fbank_once()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / reps
        gbps = fb * FBANK_BYTES_PER_UTT / (us * 1e-6) / 1e9
        result["roofline_fbank"] = {
            "kernel": "fbank_kernel<int16, unfused> (256 x 30 s -> [256, 2998, 80] fp32 log-mel)",
            "bound": "hbm",
            "achieved": gbps,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": gbps / HBM_PEAK_GBPS,
            "traffic": hbm_traffic("fbank") if fb == 256 else None,
            "launch_us": us,
            "bytes_per_launch": fb * FBANK_BYTES_PER_UTT,
            "utt_per_s": fb / (us * 1e-6),
        }
        fu = fbank_units() if fb == 256 else None
        if fu is not None:
            # the kernel is bound by the LDS and the vector ALU, not by HBM: what it could reach with its busiest unit
            # saturated, as it is and with the mel gather's bank conflicts gone (profile-time fractions, this run's rate)
            result["roofline_fbank"]["practical_ceiling"] = {
                "lds_busy_frac": fu["lds_busy_frac"], "lds_bank_conflict_frac": fu["lds_bank_conflict_frac"],
                "valu_busy_frac": fu["valu_busy_frac"],
                "ceiling_gbps_as_is": gbps * fu["speedup_bound_as_is"],
                "ceiling_gbps_without_bank_conflicts": gbps * fu["speedup_bound_without_bank_conflicts"],
                "ceiling_frac_of_hbm_peak": gbps * fu["speedup_bound_without_bank_conflicts"] / HBM_PEAK_GBPS,
                "source": fu["file"]}
        del pcm_f, logmel

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(sd, args.cpu_clips)

    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
