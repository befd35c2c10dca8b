#!/bin/bash
# round 4, fourth GPU call: blur_small with vector-load prefetch, equal row shares in k_tile, compositor occupancy sweep
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
timeout -k 10 900 python -m pytest tests/test_gpu_properties.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -m gpu -q > $OUT/r04e_tests1.log 2>&1; echo "tests1 exit $?"; tail -5 $OUT/r04e_tests1.log
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -m gpu -q -k "kitti_25 or cityscapes_half or nuscenes_5 or nuscenes_100" > $OUT/r04e_tests2.log 2>&1; echo "tests2 exit $?"; tail -5 $OUT/r04e_tests2.log
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --sweep "12=0" --sweep "6=4" --sweep "6=5" > $OUT/r04e_sweep.out 2> $OUT/r04e_sweep.err; echo "sweep exit $?"; grep SWEEP $OUT/r04e_sweep.err | cut -c1-600
timeout -k 10 600 scripts/phase_timing.sh r04e; echo "phases exit $?"
