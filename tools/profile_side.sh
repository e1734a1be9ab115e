#!/bin/bash
# Kernel-trace statistics of the decoder head and of the vocoder at their bench shapes; usage: profile_side.sh <tag>
set -e
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/side_$1
mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o dec -- python3 $GRAFT_REPO_ROOT/tools/decoder_bench.py --iters 10 > $OUT/decoder_bench.txt 2> $OUT/dec.log
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o voc -- python3 $GRAFT_REPO_ROOT/tools/vocoder_bench.py --iters 10 > $OUT/vocoder_bench.txt 2> $OUT/voc.log
# HBM traffic of the same two programs: one counter per run, 3 warm-up passes + 1 (the JSON maker divides by 4)
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o dec_f -- python3 $GRAFT_REPO_ROOT/tools/decoder_bench.py --iters 2 > $OUT/dec_f.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o dec_w -- python3 $GRAFT_REPO_ROOT/tools/decoder_bench.py --iters 2 > $OUT/dec_w.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o voc_f -- python3 $GRAFT_REPO_ROOT/tools/vocoder_bench.py --iters 1 > $OUT/voc_f.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o voc_w -- python3 $GRAFT_REPO_ROOT/tools/vocoder_bench.py --iters 1 > $OUT/voc_w.log 2>&1
ls $OUT
