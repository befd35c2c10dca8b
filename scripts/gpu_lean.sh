#!/bin/bash
# Lean headline bench (kernel times only) + optional option sweeps: scripts/gpu_lean.sh <tag> ["sweep spec" ...]
TAG=${1:-lean}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO
SW=""; for s in "$@"; do SW="$SW --sweep $s"; done
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --sweep "1=1" $SW > $OUT/${TAG}_sweep.out 2> $OUT/${TAG}_sweep.err; echo "sweep exit $?"; grep SWEEP $OUT/${TAG}_sweep.err | cut -c1-460
