#!/bin/bash
# Run on the GPU box (via gpurun): the driver tests, then the driver end to end with the host's deflate and with the device's.
# Usage: scripts/gpu_e2e_ab.sh <tag>
TAG=${1:-e2e}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
timeout -k 10 900 python -m pytest tests/test_gpu_driver.py tests/test_gpu_pipeline_async.py -m gpu -q > $OUT/${TAG}_tests.log 2>&1; echo "tests exit $?"; tail -8 $OUT/${TAG}_tests.log
for DEV in 0 1; do
  RAIN_PNG_DEVICE=$DEV timeout -k 10 500 python scripts/driver_e2e.py --frames 1024 --batch 128 2> $OUT/${TAG}_e2e_dev$DEV.err | tail -1 > $OUT/${TAG}_e2e_dev$DEV.json; echo "e2e device=$DEV exit $?"; cut -c1-700 $OUT/${TAG}_e2e_dev$DEV.json
done
