#!/bin/bash
set -e
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm3
mkdir -p $OUT
python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py --check --iters 30 --flags 0 16 32 > $OUT/time.log 2>&1
for fl in 16 32; do
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o f$fl -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py --iters 2 --flags $fl > $OUT/f$fl.log 2>&1
done
