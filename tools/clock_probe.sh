#!/bin/bash
# Samples the shader clock and the socket power (rocm-smi) every 100 ms while a bench step loop runs, for two loads:
# the headline batch (250 workgroups of the dominant kernel: every CU busy) and half of it (125 workgroups: half the
# CUs idle).  Shows whether the chip holds its clock under the load.  usage: clock_probe.sh <out-file>
out=${1:-gpurun_out/clock_probe.txt}
for b in 64 32; do
  echo "== batch $b" >> $out
  python3 bench.py --batch $b --streams 1 --no-cpu-baseline --no-extras --no-stages --steps 400 --warmup 3 > /tmp/cp_$b.json 2>/dev/null &
  pid=$!
  sleep 4
  for i in $(seq 1 12); do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket|mclk" | tr '\n' ' ' >> $out; echo >> $out
    sleep 0.1
  done
  wait $pid
  python3 - >> $out <<PY
import json
d=json.loads(open("/tmp/cp_$b.json").read().strip().splitlines()[-1])
print("batch", $b, "ms_per_step", d["ms_per_step"], "utt/s", d["value"])
PY
done
cat $out
