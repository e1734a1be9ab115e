#!/usr/bin/env python3
"""Per-convolution time, matrix rate and algorithmic memory rate of one HiFi-GAN forward (published V1 layout), from the
kernel trace rocprofv3 wrote for tools/vocoder_bench.py:  python tools/voc_trace.py <dir>/t_kernel_trace.csv"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "asrfe" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "mel_to_ntc" in r["Kernel_Name"]]
fw = rows[idx[-1]:]
B, T, C = 128, 94, 512
seq = [("mel", 0, 0), ("pre k7 80->512", 2 * B * T * 80 * 512 * 7, B * T * (80 * 2 + 512 * 2))]
for i, (u, k) in enumerate(((8, 16), (8, 16), (2, 4), (2, 4))):
    co = C // 2
    fl = 2 * B * T * C * co * k
    T *= u
    seq.append((f"ups{i} {C}->{co} k{k}", fl, B * T * co * 6))
    C = co
    for kk in (3, 7, 11):
        for d in (1, 3, 5):
            fl = 2 * B * T * C * C * kk
            seq.append((f"s{i} c1 k{kk} d{d}", fl, B * T * C * 4))
            seq.append((f"s{i} c2 k{kk}", fl, B * T * C * 12))
    seq.append((f"scale{i}", 0, B * T * C * 6))
seq.append(("post", 0, 0))
tot = {}
for (name, fl, by), r in zip(seq, fw):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    kn = r["Kernel_Name"].split("asrfe::")[1][:30]
    key = name.split()[0] + " " + (name.split()[1] if name.startswith("s") and len(name.split()) > 1 and name[1].isdigit() else "")
    tot[key] = tot.get(key, 0) + us
    if "-v" in sys.argv:
        print(f"{name:22s} {kn:32s} {us:8.1f} us  {fl / us / 1e6 if fl else 0:7.1f} TF  {by / us / 1e3:7.1f} GB/s")
for k, v in tot.items():
    print(f"{k:12s} {v:9.1f} us")
print("total", sum(tot.values()))
