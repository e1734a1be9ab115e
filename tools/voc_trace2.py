#!/usr/bin/env python3
"""The launches of the last HiFi-GAN forward in a rocprofv3 kernel trace of tools/vocoder_bench.py, in order, with their
durations, and the sum per kernel:  python tools/voc_trace2.py <dir>/t_kernel_trace.csv"""
import csv
import re
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "asrfe" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "mel_to_ntc" in r["Kernel_Name"]]
fw = rows[idx[-1]:]
tot = {}
for i, r in enumerate(fw):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    kn = re.sub(r"\(.*", "", r["Kernel_Name"].split("asrfe::")[1])
    tot[kn] = tot.get(kn, 0.0) + us
    if "-v" in sys.argv:
        print(f"{i:4d} {kn:44s} grid {r.get('Grid_Size_X', '?'):>9s} x {r.get('Grid_Size_Z', '?'):>6s} {us:9.1f} us")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"{k:44s} {v:9.1f} us")
print("total", sum(tot.values()), "us; wall", (int(fw[-1]["End_Timestamp"]) - int(fw[0]["Start_Timestamp"])) / 1e3, "us")
