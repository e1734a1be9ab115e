#!/usr/bin/env python3
"""Checks the hand-counted `s_waitcnt vmcnt(N)` waits of the post-attention kernel's output projection against the
order in which the compiled kernel really issues its loads.

    python tools/check_vm_queue.py [post_attn.s]       (without an argument the assembly is generated with hipcc)

The tiled variant of `post_attn_kernel` (csrc/post_attn.hip) issues its weight pieces by LDS-DMA and its residual tiles
by `global_load_dwordx4`, all through inline assembly, and waits for them with counted `vmcnt(N)`: loads return in issue
order, so "all but the N youngest have landed" names an exact point of the queue.  The counts come from a compile-time
model of that queue (`out_proj_vmcnt*` in the source).  This script replays the queue from the generated assembly:
in the middle of piece i of the pass (after 16 i + 8 MFMAs) the wait must cover the last DMA load of weight piece i + 1
and, in an even piece i >= 6, the last load of the residual tile that piece i - 6 requested; and it must cover nothing
younger than the later of the two (a stricter wait is legal but stalls on loads that are not needed yet).
Exit status 1 on any mismatch."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN5asrfe16post_attn_kernelILb1ELi2ELb1EEEvNS_9FfnParamsE"  # <fp16, qkv output, tiled residual stream>
# the same kernel on v_mfma_f32_16x16x32 (csrc/post_attn16.hip): the same queue, 32 MFMAs per piece instead of 16
KERNEL16 = "_ZN5asrfe12_GLOBAL__N_118post_attn16_kernelILb1ELi2EEEvNS_9FfnParamsE"


def generate_asm(source="post_attn.hip"):
    csrc = os.path.join(ROOT, "toolbox-for-asr-and-tts_amd", "csrc")
    out = os.path.join(tempfile.mkdtemp(), source.replace(".hip", ".s"))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                           "-I" + csrc, "-S", "--cuda-device-only", "-o", out, os.path.join(csrc, source)],
                          stderr=subprocess.DEVNULL)
    return out


def events(path, kernel=KERNEL):
    """[(kind, mfmas issued before it, N)] for the kernel: kind 'D' LDS-DMA load, 'X' register load issued through inline
    assembly, 'o' any other load, 'W' a hand-written vmcnt wait with its N."""
    lines = open(path).read().splitlines()
    st = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    end = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
    ev, inasm, mf = [], False, 0
    for l in lines[st:end]:
        t = l.strip()
        if "ASMSTART" in t:
            inasm = True
        elif "ASMEND" in t:
            inasm = False
        elif t.startswith(";"):
            continue
        elif "v_mfma" in t:
            mf += 1
        elif "global_load_lds_dwordx4" in t:
            ev.append(("D", mf, None))
        elif "global_load_dwordx4" in t and inasm:
            ev.append(("X", mf, None))
        elif re.match(r"(global|buffer|scratch|flat)_load", t):
            ev.append(("o", mf, None))
        elif "s_waitcnt" in t and "vmcnt" in t and inasm:
            ev.append(("W", mf, int(re.search(r"vmcnt\((\d+)\)", t).group(1))))
    return ev


def check(path, kernel=KERNEL, mp=16):
    """mp: MFMAs per weight piece (16 for the 32x32x16 kernel, 32 for the 16x16x32 one)."""
    ev = events(path, kernel)
    half = mp // 2
    errors = []
    ops = [e for e in ev if e[0] != "W"]           # the queue, in issue order
    pre = [e for e in ops if e[1] == 0]            # issued before the first MFMA
    if [e[0] for e in pre[-4:]] != ["D"] * 4 or sum(e[0] == "X" for e in pre) != 12:
        errors.append("prologue: expected twelve residual loads and the last weight piece's four DMA loads before the pass")
    n_pre = len(pre)
    # last operation (index into ops) of weight piece w and of the residual tile that piece j of the pass requests
    last_w, last_x = {}, {}
    for w in range(2, 8):  # pieces 2..7 are the last six groups of four DMA loads of the prologue, in stream order
        last_w[w] = n_pre - 1 - 4 * (7 - w)
    for j in range(32):
        # (the second half of piece j: behind its MFMAs 9 .. 16; the first half issues no loads)
        idx = [k for k, e in enumerate(ops) if mp * j + half + 1 <= e[1] <= mp * j + mp]
        d = [k for k in idx if ops[k][0] == "D"]
        x = [k for k in idx if ops[k][0] == "X"]
        want_d = 4 if j >= 1 else 0
        want_x = 4 if (j % 2 == 0 and j <= 24) else 0
        if len(d) != want_d or len(x) != want_x:
            errors.append(f"piece {j}: {len(d)} DMA loads and {len(x)} residual loads in its second half, expected {want_d} and {want_x}")
        if d:
            last_w[j + 7] = d[-1]
        if x:
            last_x[j] = x[-1]
    # the wait in the middle of piece i: the hand-written wait that follows MFMA 16 i + 8 and precedes MFMA 16 i + 9
    pos = 0  # operations issued so far
    k = 0
    waits = {}
    for e in ev:
        if e[0] == "W":
            if e[1] >= half and (e[1] - half) % mp == 0 and e[1] < 32 * mp:
                waits[(e[1] - half) // mp] = (pos, e[2])
        else:
            pos += 1
    for i in range(32):
        if i not in waits:
            errors.append(f"piece {i}: no hand-written wait in its middle")
            continue
        issued, n = waits[i]
        complete = issued - n  # operations 0 .. complete - 1 have landed
        need = []
        if i + 1 >= 2:
            need.append(last_w[i + 1])
        if i % 2 == 0 and i >= 6:
            need.append(last_x[i - 6])
        if not need:
            continue
        hi = max(need)
        if complete <= hi:
            errors.append(f"piece {i}: vmcnt({n}) leaves operation {hi} in flight, which the piece needs")
        elif complete > hi + 1:
            errors.append(f"piece {i}: vmcnt({n}) also waits for {complete - hi - 1} load(s) younger than any it needs")
    return errors


def check_m0(path):
    """Every LDS-DMA load of every kernel in the assembly file: the nearest M0 write in front of it is a scalar move in
    the same straight-line stretch, with at least one instruction between the two (on gfx9 an LDS-DMA instruction issued
    in the cycle after a scalar M0 write may still see the old M0; the compiler pads code it generates itself, never an
    inline-assembly block).  Returns a list of error strings."""
    errors = []
    kernel, last_m0, since = None, None, 0
    n_dma = 0
    for ln, l in enumerate(open(path).read().splitlines(), 1):
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        if re.match(r"^[A-Za-z_.$][\w.$]*:", t):  # a label: control flow may join here
            if not t.startswith(".LBB") and not t.startswith("BB"):
                kernel = t[:-1]
            last_m0 = None
            continue
        op = t.split()[0]
        if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            since += 1
            continue
        if "global_load_lds" in op or (op.startswith("buffer_load") and " lds" in t):
            n_dma += 1
            if last_m0 is None:
                errors.append(f"{path}:{ln} ({kernel}): LDS-DMA load with no M0 write in its straight-line stretch")
            elif since < 1:
                errors.append(f"{path}:{ln} ({kernel}): LDS-DMA load directly behind the M0 write of line {last_m0}")
            since += 1
            continue
        ops = t[len(op):].split(",")
        if ops and ops[0].strip() == "m0" and not op.startswith(("s_cmp", "s_bitcmp")):
            if op != "s_mov_b32":
                errors.append(f"{path}:{ln} ({kernel}): M0 written by {op}")
            last_m0, since = ln, 0
            continue
        since += 1
    if n_dma == 0:
        errors.append(f"{path}: no LDS-DMA load found")
    return errors


def main():
    errors = []
    if len(sys.argv) > 1:
        errors = check(sys.argv[1])
    else:
        errors = check(generate_asm()) + ["16x16x32 kernel: " + e for e in check(generate_asm("post_attn16.hip"), KERNEL16, 32)]
    for e in errors:
        print("MISMATCH:", e)
    print("post-attention output projection: 32 mid-piece waits checked against the issue order," if not errors else "",
          "ok" if not errors else f"{len(errors)} mismatch(es)")
    sys.exit(1 if errors else 0)


if __name__ == "__main__":
    main()
