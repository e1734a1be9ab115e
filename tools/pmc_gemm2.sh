#!/bin/bash
set -e
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm2
mkdir -p $OUT
for fl in 0 1; do
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o f$fl -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py --iters 2 --flags $fl > $OUT/f$fl.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT -o h$fl -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py --iters 2 --flags $fl > $OUT/h$fl.log 2>&1
done
