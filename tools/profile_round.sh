#!/bin/bash
# Kernel-trace statistics and PMC passes of one default bench step; usage: profile_round.sh <tag>
# (every rocprofv3 run is its own process with the program directly behind "--"; counters never share a run
# with a trace)
set -e
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$1
mkdir -p $OUT
B="$GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --graph 0"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 $B --steps 10 --warmup 3 > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
# the same with one stream per GPU: whole-batch launches, the shape the per-kernel roofline figures are quoted for
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace1 -- python3 $B --steps 10 --warmup 3 --streams 1 > $OUT/bench_streams1_under_rocprof.json 2> $OUT/trace1.log
P="$GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-stages --no-cpu-baseline --no-extras --streams 1 --graph 0"
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o p3 -- python3 $P > $OUT/p3.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT -o p4 -- python3 $P > $OUT/p4.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT -o p1 -- python3 $P > $OUT/p1.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p2 -- python3 $P > $OUT/p2.log 2>&1
ls $OUT
