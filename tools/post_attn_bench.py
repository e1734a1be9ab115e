#!/usr/bin/env python3
"""Times and checks the fused post-attention kernel in isolation (through asrfe_debug_post_attn).

    python tools/post_attn_bench.py [--check] [--qkv] [--iters 30] [--flags 0 1 2 ...]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toolbox_for_asr_and_tts_amd as pkg  # noqa: E402
from toolbox_for_asr_and_tts_amd.synth import encoder_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--rows", type=int, default=32000)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--flags", type=int, nargs="*", default=[0])
    ap.add_argument("--qkv", action="store_true", help="time the variant that also projects the next layer's q, k, v")
    ap.add_argument("--layers", type=int, default=2, help="encoder blocks to load; the timed launches walk through them "
                    "(50: every launch meets weights that no cache holds, as inside the encoder)")
    ap.add_argument("--tiled", action="store_true", help="residual stream in the kernel's register order (flags bit 9)")
    ap.add_argument("--lib", default=None, help="path of another build of libasrfe.so (A/B runs of experiment builds)")
    args = ap.parse_args()
    if args.lib:
        pkg._lib.LIB_PATH = os.path.abspath(args.lib)
    enc = pkg.SANMEncoder(num_blocks=args.layers)
    sd = encoder_state_dict(num_blocks=args.layers, seed=3)
    gen = torch.Generator().manual_seed(5)
    for k in sd:  # non-trivial LayerNorm parameters and biases, so that a column mix-up shows
        if "norm" in k or k.endswith(".bias"):
            sd[k] = sd[k] + 0.1 * torch.randn(sd[k].shape, generator=gen)
    enc.load_state_dict(sd)
    ctx, lib = enc.ctx, enc.ctx.lib
    dev = "cuda"
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def tens(m):
        att = (torch.randn(m, 512, device=dev) * 0.5).half()
        fsmn = (torch.randn(m, 512, device=dev) * 0.5).half()
        return att, fsmn

    TILED = 512 if args.tiled else 0

    def to_tiled(x):
        """[m][512] row-major -> the kernel's register order, padded to whole 128-row tiles: per 32-row group
        [tile nt 16][quad g 4][h 2][row 32][4 floats]."""
        m = x.shape[0]
        mp = (m + 127) // 128 * 128
        xp = torch.zeros(mp, 512, device=x.device, dtype=x.dtype)
        xp[:m] = x
        return xp.view(mp // 32, 32, 16, 4, 2, 4).permute(0, 2, 3, 4, 1, 5).contiguous().view(mp, 512)

    def from_tiled(xt, m):
        mp = xt.shape[0]
        return xt.view(mp // 32, 16, 4, 2, 32, 4).permute(0, 4, 1, 2, 3, 5).contiguous().view(mp, 512)[:m]

    if args.check:
        g = {k: v.to(dev).float() for k, v in sd.items()}
        pfx = "encoders0.0."
        for m in (1, 33, 128, 1000):
            for with_x in (True, False):
                x = torch.randn(m, 512, device=dev) * 2.0 + 0.3
                att, fsmn = tens(m)
                xo = torch.empty_like(x)
                y = torch.empty(m, 512, device=dev, dtype=torch.float16)
                xin = to_tiled(x) if args.tiled else x
                if args.tiled:  # x_out alone: tiled in, row-major out (the encoder's last layer); no y in this mode
                    ctx.check(lib.asrfe_debug_post_attn(ctx.handle, 0, att.data_ptr(), fsmn.data_ptr(),
                                                        xin.data_ptr() if with_x else None, xo.data_ptr(), None, None, m, TILED, stream))
                    y = None
                else:
                    ctx.check(lib.asrfe_debug_post_attn(ctx.handle, 0, att.data_ptr(), fsmn.data_ptr(),
                                                        x.data_ptr() if with_x else None, xo.data_ptr(), y.data_ptr(), None, m, 0, stream))
                xo2 = torch.full(((m + 127) // 128 * 128, 512), float("nan"), device=dev)
                qkv = torch.empty(m, 1536, device=dev, dtype=torch.float16)
                ctx.check(lib.asrfe_debug_post_attn(ctx.handle, 0, att.data_ptr(), fsmn.data_ptr(),
                                                    xin.data_ptr() if with_x else None, xo2.data_ptr(), None, qkv.data_ptr(), m, TILED, stream))
                xo2 = from_tiled(xo2, m) if args.tiled else xo2[:m]
                xm = att.float() @ g[pfx + "self_attn.linear_out.weight"].half().float().T + g[pfx + "self_attn.linear_out.bias"] + fsmn.float()
                if with_x:
                    xm = xm + x
                h = torch.nn.functional.layer_norm(xm, (512,), g[pfx + "norm2.weight"], g[pfx + "norm2.bias"], 1e-12)
                h = torch.relu(h.half().float() @ g[pfx + "feed_forward.w_1.weight"].half().float().T + g[pfx + "feed_forward.w_1.bias"])
                ref = xm + h.half().float() @ g[pfx + "feed_forward.w_2.weight"].half().float().T + g[pfx + "feed_forward.w_2.bias"]
                yref = torch.nn.functional.layer_norm(ref, (512,), None, None, 1e-12)  # scale / shift live in the next QKV
                n1 = torch.nn.functional.layer_norm(ref, (512,), g["encoders.0.norm1.weight"], g["encoders.0.norm1.bias"], 1e-12)
                qref = n1 @ g["encoders.0.self_attn.linear_q_k_v.weight"].T + g["encoders.0.self_attn.linear_q_k_v.bias"]
                e1 = (xo - ref).abs().max().item()
                e2 = (y.float() - yref).abs().max().item() if y is not None else 0.0
                e3 = (qkv.float() - qref).abs().max().item()
                print(f"check m={m} x={'yes' if with_x else 'no '}: x_out err {e1:.3e}, y err {e2:.3e}, qkv err {e3:.3e}, "
                      f"x_out of both variants equal: {torch.equal(xo, xo2)}")
                assert e1 < 3e-3 and e2 < 1e-2 and e3 < 2e-2 and torch.equal(xo, xo2)
    m = args.rows
    x = torch.randn((m + 127) // 128 * 128, 512, device=dev)
    y = torch.empty(m, 512, device=dev, dtype=torch.float16)
    att, fsmn = tens(m)

    qkv = torch.empty(m, 1536, device=dev, dtype=torch.float16)

    walk = [0]

    def launch(flags):
        walk[0] = (walk[0] + 1) % (args.layers - 1)
        ctx.check(lib.asrfe_debug_post_attn(ctx.handle, walk[0], att.data_ptr(), fsmn.data_ptr(), x.data_ptr(), x.data_ptr(),
                                            None if (args.qkv or args.tiled) else y.data_ptr(), qkv.data_ptr() if args.qkv else None,
                                            m, flags | TILED, stream))

    import time
    t0 = time.time()
    while time.time() - t0 < 1.5:  # let the clocks settle before the first measurement
        for _ in range(50):
            launch(0)
        x.normal_()
        torch.cuda.synchronize()
    # clock reference: boxes differ by up to ~10 %, so every run also times the plain GEMM (FFN1 shape)
    ga = (torch.randn(m, 512, device=dev) * 0.5).half()
    gw = (torch.randn(2048, 512, device=dev) * 0.05).half()
    gb = torch.randn(2048, device=dev)
    go = torch.empty(m, 2048, device=dev, dtype=torch.float16)

    def ref_gemm():
        for _ in range(10):
            ctx.check(lib.asrfe_debug_gemm(ctx.handle, ga.data_ptr(), 512, gw.data_ptr(), 512, gb.data_ptr(), m, 2048, 512,
                                           go.data_ptr(), 2048, None, None, 1, 0, stream))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            ctx.check(lib.asrfe_debug_gemm(ctx.handle, ga.data_ptr(), 512, gw.data_ptr(), 512, gb.data_ptr(), m, 2048, 512,
                                           go.data_ptr(), 2048, None, None, 1, 0, stream))
        e1.record()
        torch.cuda.synchronize()
        print(f"reference gemm [{m} x 512] x [512 x 2048]: {1e3 * e0.elapsed_time(e1) / args.iters:7.1f} us")

    if any(f & 256 for f in args.flags):  # timing build: per-phase cycle stamps of one wave
        for fl in [f for f in args.flags if f & 256]:
            tr = torch.zeros(64, dtype=torch.int64, device=dev)
            for _ in range(3):
                ctx.check(lib.asrfe_debug_post_attn(ctx.handle, 0, att.data_ptr(), fsmn.data_ptr(), x.data_ptr(), x.data_ptr(),
                                                    tr.data_ptr(), qkv.data_ptr(), m, fl | TILED, stream))
            torch.cuda.synchronize()
            t = [v for v in tr.cpu().tolist() if v]
            names = ["prologue", "out-proj", "LayerNorm", "feed-forward", "x_out + fragments", "q pass", "k pass", "v pass", "stores"]
            print(f"flags {fl} phase cycles:", {n: b - a for n, a, b in zip(names, t, t[1:])}, "total", t[-1] - t[0])
        return
    ref_gemm()
    for flags in args.flags * args.repeat:
        def run():
            launch(flags)
        x.normal_()
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / args.iters
        flop = 2.0 * m * (2048 * 512 * 2 + 512 * 512 + (1536 * 512 if args.qkv else 0))
        print(f"post_attn{'+qkv' if args.qkv else ''} M={m} flags={flags}: {us:7.1f} us  {flop / us / 1e6:7.1f} TFLOP/s")
    ref_gemm()


if __name__ == "__main__":
    main()
