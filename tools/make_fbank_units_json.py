#!/usr/bin/env python3
"""Builds profiles/<tag>_fbank_units.json from the two SQ counter passes of tools/pmc_fbank.sh over the fbank-only kernel
(256 x 30 s): what fraction of the kernel's cycles the LDS and the vector ALU are busy, and what the kernel could reach
with those units saturated.

    python tools/make_fbank_units_json.py <tag> <pmc dir>

SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT are cycles summed over the 256 CUs; SQ_ACTIVE_INST_VALU counts units of four
cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.buildtag import ROOT, csrc_tag  # noqa: E402
from tools.make_traffic_json import means  # noqa: E402


def main():
    tag, d = sys.argv[1], sys.argv[2]
    m = means(d)
    c = next(v for k, v in m.items() if "fbank_kernel<short, false" in k)
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    lds = c["SQ_LDS_IDX_ACTIVE"] / 256.0 / cyc
    conflict = c["SQ_LDS_BANK_CONFLICT"] / 256.0 / cyc
    valu = c["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc
    out = {"source": "rocprofv3 --pmc, two SQ passes over tools/fbank_bench.py (tools/pmc_fbank.sh), fbank-only kernel, 256 x 30 s",
           "csrc_tag": csrc_tag(), "kernel_cycles": cyc,
           "lds_busy_frac": round(lds, 4), "lds_bank_conflict_frac": round(conflict, 4), "valu_busy_frac": round(valu, 4),
           # wave-level instructions per group of 4 frames (one wave carries 4 frames at a time)
           "insts_per_4_frames": {"valu": round(c["SQ_INSTS_VALU"] / (256 * 2998.0 / 4), 1), "lds": round(c["SQ_INSTS_LDS"] / (256 * 2998.0 / 4), 1)},
           # the kernel cannot run faster than its busiest unit allows
           "speedup_bound_as_is": round(1.0 / max(lds, valu), 3),
           "speedup_bound_without_bank_conflicts": round(1.0 / max(lds - conflict, valu), 3)}
    path = os.path.join(ROOT, "profiles", f"{tag}_fbank_units.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
