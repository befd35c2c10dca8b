#!/bin/bash
# GPU box: a parity subset with the product library, then one bench sweep per library (the product's and build_alt/ ones).
# Usage: scripts/gpu_ab_libs.sh <tag> "<pytest selection or 'none'>" "<sweep args>" [alt tags ...]
TAG=$1; SEL=$2; SW=$3; shift 3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
if [ "$SEL" != "none" ]; then timeout -k 10 1500 python -m pytest $SEL -m gpu -q -x > $OUT/${TAG}_tests.log 2>&1; echo "tests exit $?"; tail -3 $OUT/${TAG}_tests.log; fi
run() { timeout -k 10 600 python bench.py --steps 5 --warmup 2 --sweep "1=1" $SW > $OUT/${TAG}_$1.out 2> $OUT/${TAG}_$1.err; echo "[$1] exit $?"; grep '^SWEEP' $OUT/${TAG}_$1.err | cut -c1-700; }
run product
for A in "$@"; do RAINHIP_LIB=$REPO/build_alt/librainhip_$A.so run $A; done
