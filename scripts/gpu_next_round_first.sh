#!/bin/bash
# First GPU call of the next round: everything needed to decide on RR_OPT_FOV_F32 (float32 environment-map sums, measured
# at k_fov_sums 3.3 -> 2.1 ms in round 3 and left off) in one go.
#   1. the whole GPU tier with the option in its "auto" mode (float sums whenever the compositor blends float colours);
#      failures here are the tests whose tolerance has to be stated at image level before the default can change;
#   2. the headline loop under the option and under the workgroup shapes the float kernel makes possible (half the
#      registers: 512-thread workgroups, two per CU);
#   3. the whole GPU tier with the defaults (the state of the repository).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
TAG=${1:-r04a}
cd $REPO
RAINHIP_OPTIONS=10=2 timeout -k 10 420 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/${TAG}_gpu_tests_fov_f32.log 2>&1
echo "tests with 10=2: exit $?"; tail -15 $OUT/${TAG}_gpu_tests_fov_f32.log | cut -c1-200
# option ids (include/rainhip.h): 3 = FOV threads (512 | 1024), 4 = FOV drops per thread (1 | 2 | 4 | 8), 10 = float32 sums
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --sweep "10=2" --sweep "10=2,3=512,4=8" --sweep "10=2,3=512,4=4" --sweep "10=2,3=1024,4=4" \
  > $OUT/${TAG}_sweep.out 2> $OUT/${TAG}_sweep.err
echo "sweep exit $?"; grep SWEEP $OUT/${TAG}_sweep.err | cut -c1-330
timeout -k 10 420 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/${TAG}_gpu_tests.log 2>&1
echo "tests (defaults): exit $?"; tail -3 $OUT/${TAG}_gpu_tests.log
