#!/bin/bash
# PMC passes over tools/gemm_bench.py (each pass its own run; SQ block holds 8 counters)
set -e
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT -o p1 -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py --iters 3 --flags 0 > $OUT/p1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT -o p2 -- python3 $GRAFT_REPO_ROOT/tools/gemm_bench.py --iters 3 --flags 0 > $OUT/p2.log 2>&1
ls $OUT
