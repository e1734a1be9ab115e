#!/usr/bin/env python3
"""Times and checks the encoder GEMM kernel in isolation (through asrfe_debug_gemm).

    python tools/gemm_bench.py [--check] [--iters 50]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import toolbox_for_asr_and_tts_amd as pkg  # noqa: E402

SHAPES = [  # name, M, N, K, epi, add16, add32
    ("qkv", 32000, 1536, 512, 0, False, False),
    ("out", 32000, 512, 512, 2, True, True),
    ("ffn1", 32000, 2048, 512, 1, False, False),
    ("ffn2", 32000, 512, 2048, 2, False, True),
    # the decoder's products over 64 x 256 token rows: 128 full-height tiles for 256 CUs (flags 256 / 512 force a height)
    ("dq", 16384, 512, 512, 0, False, False),
    ("do", 16384, 512, 512, 2, False, True),
    ("dw2", 16384, 512, 2048, 2, False, True),
    ("dw1", 16384, 2048, 512, 1, False, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--flags", type=int, nargs="*", default=[0, 1])
    ap.add_argument("--lib", default=None, help="path of another build of libasrfe.so (A/B timing)")
    args = ap.parse_args()
    if args.lib:
        pkg._lib.LIB_PATH = os.path.abspath(args.lib)
    ctx = pkg.Context()
    lib = ctx.lib
    dev = "cuda"
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if args.check:
        for (m, n, k, epi) in [(300, 256, 128, 0), (513, 512, 256, 1), (1000, 512, 384, 2), (777, 768, 640, 2), (256, 256, 2048, 0)]:
            a = (torch.randn(m, k, device=dev) * 0.5).half()
            w = (torch.randn(n, k, device=dev) * 0.05).half()
            bias = torch.randn(n, device=dev)
            add16 = torch.randn(m, n, device=dev).half()
            add32 = torch.randn(m, n, device=dev)
            out = torch.empty(m, n, device=dev, dtype=torch.float32 if epi == 2 else torch.float16)
            ctx.check(lib.asrfe_debug_gemm(ctx.handle, a.data_ptr(), k, w.data_ptr(), k, bias.data_ptr(), m, n, k,
                                           out.data_ptr(), n, add16.data_ptr() if epi == 2 else None,
                                           add32.data_ptr() if epi == 2 else None, epi, 0, stream))
            ref = a.float() @ w.float().T + bias
            if epi == 1:
                ref = torch.relu(ref)
            if epi == 2:
                ref = ref + add16.float() + add32
            err = (out.float() - ref).abs().max().item()
            tol = 2e-2 if epi != 2 else 2e-3
            print(f"check M={m} N={n} K={k} epi={epi}: max err {err:.3e}", "OK" if err < tol else "FAIL")
            assert err < tol
    for name, m, n, k, epi, u16, u32 in SHAPES:
        a = (torch.randn(m, k, device=dev) * 0.5).half()
        w = (torch.randn(n, k, device=dev) * 0.05).half()
        bias = torch.randn(n, device=dev)
        add16 = torch.randn(m, n, device=dev).half() if u16 else None
        add32 = torch.randn(m, n, device=dev) if u32 else None
        out = add32 if (epi == 2 and u32) else torch.empty(m, n, device=dev, dtype=torch.float32 if epi == 2 else torch.float16)
        for flags in args.flags:
            def run():
                ctx.check(lib.asrfe_debug_gemm(ctx.handle, a.data_ptr(), k, w.data_ptr(), k, bias.data_ptr(), m, n, k,
                                               out.data_ptr(), n, add16.data_ptr() if u16 else None,
                                               add32.data_ptr() if u32 else None, epi, flags, stream))
            for _ in range(5):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / args.iters
            print(f"{name:5s} M={m} N={n} K={k} flags={flags}: {us:7.1f} us  {2.0 * m * n * k / us / 1e6:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
