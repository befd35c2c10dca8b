#!/bin/bash
TAG=${1:-r03l}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO
timeout -k 10 1200 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_tests.log 2>&1; echo "tests exit $?"; tail -6 $OUT/${TAG}_tests.log
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --sweep "3=512,4=4" > $OUT/${TAG}_sweep.out 2> $OUT/${TAG}_sweep.err; echo "sweep exit $?"; grep SWEEP $OUT/${TAG}_sweep.err | cut -c1-420
