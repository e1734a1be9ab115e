#!/bin/bash
# HBM traffic counters of the fbank micro-benchmark (one run per counter set); usage: pmc_fbank_traffic.sh <tag>
set -e
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
mkdir -p $OUT
ARGS="$GRAFT_REPO_ROOT/tools/fbank_bench.py --iters 2 --repeat 1"
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o p3 -- python3 $ARGS > $OUT/p3.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT -o p4 -- python3 $ARGS > $OUT/p4.log 2>&1
ls $OUT
