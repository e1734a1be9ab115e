#!/usr/bin/env python3
"""Times and checks the attention + FSMN kernel in isolation (through asrfe_debug_attention).

    python tools/attn_bench.py [--check] [--iters 30] [--flags 0 1 ...]
"""
import argparse
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toolbox_for_asr_and_tts_amd as pkg  # noqa: E402
from toolbox_for_asr_and_tts_amd.synth import encoder_state_dict  # noqa: E402


def reference(qkv, lens, w):
    """fp64 attention and FSMN on the 16-bit-rounded inputs. qkv [B, T, 1536], w [512, 1, 11]."""
    B, T, _ = qkv.shape
    q, k, v = [t.double() for t in qkv.split(512, dim=-1)]
    mask = (torch.arange(T, device=qkv.device)[None, :] < lens[:, None])
    att = torch.zeros(B, T, 512, dtype=torch.float64, device=qkv.device)
    for hd in range(4):
        sl = slice(128 * hd, 128 * hd + 128)
        s = q[..., sl] @ k[..., sl].transpose(1, 2) / 128 ** 0.5
        s = s.masked_fill(~mask[:, None, :], float("-inf"))
        att[..., sl] = torch.softmax(s, dim=-1) @ v[..., sl]
    vm = v * mask[..., None]
    x = torch.nn.functional.pad(vm.transpose(1, 2), (5, 5))
    f = torch.nn.functional.conv1d(x, w.double(), groups=512).transpose(1, 2) + vm
    return att, f * mask[..., None]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--flags", type=int, nargs="*", default=[0])
    ap.add_argument("--batch", type=int, default=64, help="utterances of the timed launch")
    ap.add_argument("--lib", default=None, help="path of another build of libasrfe.so (A/B runs of experiment builds)")
    ap.add_argument("--len", type=int, default=500, help="valid length of every utterance (T stays 500)")
    args = ap.parse_args()
    if args.lib:
        pkg._lib.LIB_PATH = os.path.abspath(args.lib)
    enc = pkg.SANMEncoder(num_blocks=1)
    sd = encoder_state_dict(num_blocks=1, seed=3)
    sd["encoders0.0.self_attn.fsmn_block.weight"] = torch.randn(512, 1, 11) * 0.2
    enc.load_state_dict(sd)
    ctx, lib = enc.ctx, enc.ctx.lib
    dev = "cuda"
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    w = sd["encoders0.0.self_attn.fsmn_block.weight"].to(dev)

    def launch(qkv, lens, att, fsmn, flags=0):
        B, T, _ = qkv.shape
        ctx.check(lib.asrfe_debug_attention(ctx.handle, 0, qkv.data_ptr(), lens.data_ptr(), B, T, att.data_ptr(),
                                            fsmn.data_ptr(), flags, stream))

    if args.check:
        for T, ll in [(1, [1]), (7, [7, 3]), (64, [64, 5]), (128, [128, 0]), (129, [129, 1, 0]), (130, [130, 77, 128]),
                      (256, [256, 200, 64]), (257, [257, 193]), (300, [300, 65, 64, 129]), (500, [500, 333, 64, 6]),
                      (513, [513, 500, 449])]:
            B = len(ll)
            qkv = (torch.randn(B, T, 1536, device=dev) * 1.5).half()
            lens = torch.tensor(ll, dtype=torch.int32, device=dev)
            att = torch.empty(B, T, 512, device=dev, dtype=torch.float16)
            fsmn = torch.empty(B, T, 512, device=dev, dtype=torch.float16)
            launch(qkv, lens, att, fsmn)
            ra, rf = reference(qkv, lens, w)
            valid = (torch.arange(T, device=dev)[None, :] < lens[:, None])[..., None]
            # (rows past an utterance's length have no defined reference: all their keys are masked)
            e1 = torch.where(valid, att.double() - ra, torch.zeros_like(ra)).abs().max().item()
            assert torch.isfinite(att.float()).all()
            e2 = (fsmn.double() - rf).abs().max().item()
            print(f"check T={T} lens={ll}: att err {e1:.3e}, fsmn err {e2:.3e}")
            assert e1 < 4e-3 and e2 < 8e-3
    B, T = args.batch, 500
    qkv = (torch.randn(B, T, 1536, device=dev) * 1.5).half()
    lens = torch.full((B,), min(T, args.len), dtype=torch.int32, device=dev)
    att = torch.empty(B, T, 512, device=dev, dtype=torch.float16)
    fsmn = torch.empty(B, T, 512, device=dev, dtype=torch.float16)
    t0 = time.time()
    while time.time() - t0 < 1.0:
        for _ in range(50):
            launch(qkv, lens, att, fsmn)
        torch.cuda.synchronize()
    if any(f & 1024 for f in args.flags):  # timing build: the 256-row kernel's stamps, group A and group B
        fsmn.zero_()
        launch(qkv, lens, att, fsmn, 512 + 1024)
        torch.cuda.synchronize()
        raw = fsmn.view(-1)[:4 * 128].view(torch.int64).cpu().tolist()
        for name, tr in (("A", raw[:64]), ("B", raw[64:128])):
            tr = [t for t in tr if t]
            print(f"group {name} trace deltas (cycles):", [b - a for a, b in zip(tr, tr[1:])])
        return
    if any(f & 512 for f in args.flags):
        fsmn.zero_()
        for fl in args.flags:
            fsmn.zero_()
            launch(qkv, lens, att, fsmn, (fl | 1) if os.environ.get("ASRFE_ATTN_NARROW") == "1" else fl)
            torch.cuda.synchronize()
            tr = fsmn.view(-1)[:4 * 128].view(torch.int64).cpu().tolist()
            tr = [t for t in tr if t]
            print(f"flags {fl} trace deltas (cycles):", [b - a for a, b in zip(tr, tr[1:])])
        return
    # clock reference: boxes differ by up to ~10 %, so every run also times a plain GEMM of the library
    ga = (torch.randn(32000, 512, device=dev) * 0.5).half()
    gw = (torch.randn(2048, 512, device=dev) * 0.05).half()
    gb = torch.randn(2048, device=dev)
    go = torch.empty(32000, 2048, device=dev, dtype=torch.float16)

    def ref_gemm():
        def one():
            ctx.check(lib.asrfe_debug_gemm(ctx.handle, ga.data_ptr(), 512, gw.data_ptr(), 512, gb.data_ptr(), 32000, 2048,
                                           512, go.data_ptr(), 2048, None, None, 1, 0, stream))
        for _ in range(10):
            one()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            one()
        e1.record()
        torch.cuda.synchronize()
        print(f"reference gemm [32000 x 512] x [512 x 2048]: {1e3 * e0.elapsed_time(e1) / args.iters:7.1f} us")

    ref_gemm()
    for flags in args.flags * args.repeat:
        for _ in range(5):
            launch(qkv, lens, att, fsmn, flags)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            launch(qkv, lens, att, fsmn, flags)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / args.iters
        print(f"attention+fsmn B={B} T={T} flags={flags}: {us:7.1f} us  {4 * B * 4 * T * T * 128 / us / 1e6:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
