#!/bin/bash
# round 4, second GPU call: whole GPU tier (all failures shown), fov-sum option sweep on the float path, phase clocks
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
timeout -k 10 1500 python -m pytest tests -m gpu -q --durations=8 > $OUT/r04b_tests.log 2>&1; echo "tests exit $?"; tail -30 $OUT/r04b_tests.log
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --sweep "4=4" --sweep "3=512" --sweep "3=512,4=4" --sweep "10=0" > $OUT/r04b_sweep.out 2> $OUT/r04b_sweep.err; echo "sweep exit $?"; grep SWEEP $OUT/r04b_sweep.err | cut -c1-600
timeout -k 10 600 scripts/phase_timing.sh r04b; echo "phases exit $?"
