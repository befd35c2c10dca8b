#!/bin/bash
# round 4: thread-per-drop polygons (k_fov_dda) -- parity subset, A/B against the edge-parallel kernel
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
timeout -k 10 900 python -m pytest tests/test_gpu_properties.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_single_drop_seam.py -m gpu -q > $OUT/r04f_tests1.log 2>&1; echo "tests1 exit $?"; tail -12 $OUT/r04f_tests1.log
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_particles.py -m gpu -q -k "kitti_25 or cityscapes_half or nuscenes_5 or nuscenes_100 or pipeline" > $OUT/r04f_tests2.log 2>&1; echo "tests2 exit $?"; tail -8 $OUT/r04f_tests2.log
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --sweep "12=0" > $OUT/r04f_sweep.out 2> $OUT/r04f_sweep.err; echo "sweep exit $?"; grep SWEEP $OUT/r04f_sweep.err | cut -c1-600
