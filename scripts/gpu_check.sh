#!/bin/bash
# scripts/gpu_check.sh <tag> "<pytest -k expr>": a GPU test subset + the lean headline bench
TAG=${1:-chk}; KEXPR=${2:-parity}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $OUT/${TAG}_tests.log 2>&1; echo "tests exit $?"; tail -4 $OUT/${TAG}_tests.log
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --sweep "1=1" > $OUT/${TAG}_sweep.out 2> $OUT/${TAG}_sweep.err; echo "sweep exit $?"; grep SWEEP $OUT/${TAG}_sweep.err | head -1 | cut -c1-500
