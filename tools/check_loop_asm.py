#!/usr/bin/env python3
"""Guards the software pipeline of a hand-scheduled kernel against the compiler.

    python tools/check_loop_asm.py <file.s> <kernel-name-substring>

Finds, in the AMDGPU assembly of the kernel, the innermost loop with the most MFMAs and lists what must
not be in it: scratch (spill) traffic, scalar loads (they share lgkmcnt with the LDS reads the loop counts),
and any s_waitcnt vmcnt(N) other than the kernel's own (a compiler-inserted vmcnt wait drains the LDS-DMA
queue the loop keeps full).  Prints the counts; exits 1 if scratch or scalar loads are found.
"""
import collections
import re
import sys


def main():
    path, name = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l.split(":")[0] and ":" in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    best = None
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            lo = labels[m.group(1)]
            n = sum("v_mfma" in x for x in body[lo:i])
            if best is None or n > best[0]:
                best = (n, lo, i)
    n, lo, hi = best
    loop = [l for l in body[lo:hi + 1] if not l.lstrip().startswith(";")]
    ops = collections.Counter(l.split()[0] for l in loop if l.startswith("\t") and l.split())
    waits = collections.Counter(re.findall(r"(?:vmcnt|lgkmcnt)\(\d+\)", "\n".join(loop)))
    scratch = sum(v for k, v in ops.items() if k.startswith("scratch_"))
    sload = sum(v for k, v in ops.items() if k.startswith("s_load") or k.startswith("s_buffer_load"))
    print(f"{name}: loop lines {lo}..{hi}, {n} MFMAs, {scratch} scratch ops, {sload} scalar loads, "
          f"{ops.get('v_accvgpr_read_b32', 0) + ops.get('v_accvgpr_write_b32', 0) + ops.get('v_accvgpr_mov_b32', 0)} accvgpr moves")
    print("  waits:", dict(sorted(waits.items())))
    print("  top ops:", ops.most_common(12))
    sys.exit(1 if scratch or sload else 0)


if __name__ == "__main__":
    main()
