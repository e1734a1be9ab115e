#!/usr/bin/env python3
"""Instruction mix of one kernel in a gfx950 assembly listing: asm_mix.py file.s substring-of-symbol."""
import re
import sys
from collections import Counter


def main():
    path, want = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^\w+:", l) and want in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    ins = Counter()
    for l in lines[start:end]:
        t = l.strip()
        if not l.startswith("\t") or not t or t[0] in ".;":
            continue
        ins[t.split()[0]] += 1
    tot = sum(ins.values())
    grp = lambda p: sum(v for k, v in ins.items() if k.startswith(p))
    print(f"{lines[start][:70]} total {tot} valu {grp('v_')} pk {grp('v_pk')} ds {grp('ds_')} "
          f"salu {grp('s_')} vmem {grp('global_') + grp('buffer_') + grp('scratch_')}")
    for l in lines[end:end + 60]:
        if re.search(r"\.num_vgpr|private_seg_size", l):
            print("  ", l.strip().split(".")[-1])
    print("  ", ins.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 20))


if __name__ == "__main__":
    main()
