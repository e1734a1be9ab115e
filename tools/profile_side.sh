#!/bin/bash
# Kernel-trace statistics of the decoder head and of the vocoder at their bench shapes; usage: profile_side.sh <tag>
set -e
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/side_$1
mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o dec -- python3 $GRAFT_REPO_ROOT/tools/decoder_bench.py --iters 10 > $OUT/decoder_bench.txt 2> $OUT/dec.log
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o voc -- python3 $GRAFT_REPO_ROOT/tools/vocoder_bench.py --iters 10 > $OUT/vocoder_bench.txt 2> $OUT/voc.log
ls $OUT
