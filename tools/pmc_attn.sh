#!/bin/bash
# Two SQ counter passes over the attention micro-benchmark; usage: pmc_attn.sh <tag>
set -e
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
mkdir -p $OUT
ARGS="$GRAFT_REPO_ROOT/tools/attn_bench.py --iters 3 --repeat 1"
timeout -s KILL 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT -o p1 -- python3 $ARGS > $OUT/p1.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p2 -- python3 $ARGS > $OUT/p2.log 2>&1
ls $OUT
