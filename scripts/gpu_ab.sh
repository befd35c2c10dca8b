#!/bin/bash
# Run on the GPU box (via gpurun): GPU parity tests (optionally a -k subset), then ONE bench process that times the headline
# configuration and a list of rr_set_option sets (bench.py --sweep), scene set-up paid once.
# Usage: scripts/gpu_ab.sh <tag> "<pytest -k expr or empty>" "<sweep 1>" "<sweep 2>" ...    -> gpurun_out/<tag>_{tests,sweep}.log
TAG=${1:-ab}; KEXPR=${2:-}; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO
if [ -n "$KEXPR" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $OUT/${TAG}_tests.log 2>&1
else
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_tests.log 2>&1
fi
echo "tests exit $?"; tail -5 $OUT/${TAG}_tests.log
ARGS=""
for S in "$@"; do ARGS="$ARGS --sweep $S"; done
timeout 900 python bench.py --steps 5 --warmup 2 $ARGS > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_sweep.log
grep '^SWEEP' $OUT/${TAG}_sweep.log
tail -c 600 $OUT/${TAG}_bench.json
