#!/bin/bash
# round 4, first GPU call: whole GPU tier, lean bench (float32 inputs) + the float64-input / float64-colour-branch A/B, kernel stats
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
timeout -k 10 1500 python -m pytest tests -m gpu -q -x --durations=15 > $OUT/r04a_tests.log 2>&1; echo "tests exit $?"; tail -25 $OUT/r04a_tests.log
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --sweep "10=0" --sweep "10=1" > $OUT/r04a_sweep.out 2> $OUT/r04a_sweep.err; echo "sweep exit $?"; grep SWEEP $OUT/r04a_sweep.err | cut -c1-700
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --input-dtype f64 --sweep "10=0" > $OUT/r04a_sweep64.out 2> $OUT/r04a_sweep64.err; echo "sweep64 exit $?"; grep SWEEP $OUT/r04a_sweep64.err | cut -c1-700
