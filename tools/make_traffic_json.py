#!/usr/bin/env python3
"""Builds profiles/<tag>_hbm_traffic.json from the rocprofv3 counter CSVs of tools/profile_round.sh (and, optionally,
of tools/pmc_fbank_traffic.sh): mean per launch of FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE
per kernel, with the hash of the kernel sources the run was made with (tools/buildtag.py).

    python tools/make_traffic_json.py <tag> <build description> <profile_round dir> [<fbank traffic dir> [<profile_side dir>]]

FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane
streaming loads at 64 bytes and is doubled here, WRITE_SIZE is taken as it is."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.buildtag import ROOT, csrc_tag, side_tag  # noqa: E402

# (the post-attention variant that also projects the next layer's q, k, v: 49 of a step's 50 launches)
KERNELS = {"post_attn": "post_attn16_kernel<true, 2>", "attention": "attn_kernel", "gemm_qkv": "gemm_bf16_kernel"}


def means(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


def pick(m, needle):
    for k, v in m.items():
        if needle in k:
            return v
    return None


def main():
    tag, build, d = sys.argv[1], sys.argv[2], sys.argv[3]
    m = means(d)
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE, separate passes over "
                     "one single-stream bench step (tools/profile_round.sh), mean per launch; FETCH_SIZE (KiB) doubled "
                     "(gfx950 tallies the 128-byte requests of 16-byte-per-lane streaming loads at 64 bytes); WRITE_SIZE "
                     "(KiB) as is",
           "build": build, "csrc_tag": csrc_tag(), "kernels": {}, "mfma_pipe": {}}
    for name, needle in KERNELS.items():
        c = pick(m, needle)
        if not c:
            continue
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            out["kernels"][name] = {"fetch_bytes": int(c["FETCH_SIZE"] * 1024 * 2), "write_bytes": int(c["WRITE_SIZE"] * 1024)}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            per_simd = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0
            gui = c["GRBM_GUI_ACTIVE"] / 8.0  # the counter is summed over the 8 XCDs
            out["mfma_pipe"][name] = {"mfma_busy_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"], "per_simd": per_simd,
                                      "gui_active_cycles_per_xcd": gui, "busy_frac_of_kernel_cycles": round(per_simd / gui, 4)}
    if len(sys.argv) > 4:
        mf = means(sys.argv[4])
        for k, c in mf.items():
            if "fbank_kernel" in k and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                fused = "true" in k.split("fbank_kernel<")[1].split(",")[1]
                out["kernels"]["frontend_fused" if fused else "fbank"] = {
                    "fetch_bytes": int(c["FETCH_SIZE"] * 1024 * 2), "write_bytes": int(c["WRITE_SIZE"] * 1024),
                    "note": k[:90] + ", 256 x 30 s (tools/pmc_fbank_traffic.sh)"}
    if len(sys.argv) > 5:
        # tools/profile_side.sh: whole programs (decoder head, vocoder), 4 passes each; every asrfe kernel's launches added up
        out["side"] = {}
        out["side_tag"] = side_tag()
        for prog in ("dec", "voc"):
            tot = {}
            for cname, suffix in (("FETCH_SIZE", "f"), ("WRITE_SIZE", "w")):
                s = 0.0
                for f in sorted(glob.glob(os.path.join(sys.argv[5], "**", f"{prog}_{suffix}_counter_collection.csv"), recursive=True)):
                    for r in csv.DictReader(open(f)):
                        if "asrfe" in r["Kernel_Name"] and r["Counter_Name"] == cname:
                            s += float(r["Counter_Value"])
                tot[cname] = s
            if tot["FETCH_SIZE"] > 0 and tot["WRITE_SIZE"] > 0:
                out["side"]["decoder" if prog == "dec" else "vocoder"] = {
                    "fetch_bytes": int(tot["FETCH_SIZE"] * 1024 * 2 / 4), "write_bytes": int(tot["WRITE_SIZE"] * 1024 / 4),
                    "note": "per pass of tools/%s_bench.py (all its kernels), 4 passes per counter run" % ("decoder" if prog == "dec" else "vocoder")}
    path = os.path.join(ROOT, "profiles", f"{tag}_hbm_traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
