#!/bin/bash
# GPU box: the step time of the product library and of build_alt/ ones, alternating, <reps> times (10 steps each).
# Usage: scripts/gpu_ab_rep.sh <reps> <alt tag> ...
REPS=$1; shift
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq $REPS); do
for L in product "$@"; do
  if [ $L = product ]; then unset RAINHIP_LIB; else export RAINHIP_LIB=$PWD/build_alt/librainhip_$L.so; fi
  python bench.py --steps 10 --warmup 3 --sweep "1=1" 2>&1 >/dev/null | grep SWEEP | head -1 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l[6:]); print('$L', d['ms_per_step'])"
done; done
