#!/bin/bash
# One GPU session of a build: GPU tests, the default bench line, kernel-trace statistics and the counter passes.
# usage: tools/gpu_round.sh <tag> [tests|notests]
set -e
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ "$2" != "notests" ]; then
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
  tail -3 $OUT/pytest.log
fi
timeout -k 10 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
cut -c1-600 $OUT/bench.json
timeout -k 10 600 bash tools/profile_round.sh $TAG && echo profile done
