#!/bin/bash
# Round-3 closing call: the whole GPU test tier on the final host code, the driver end to end on both host routes, and the
# host pipeline's timeline (kernel + copy trace) in the bench's configuration.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
TAG=${1:-r03k}
cd $REPO
timeout -k 10 420 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/${TAG}_gpu_tests.log 2>&1
echo "tests exit $?"; tail -3 $OUT/${TAG}_gpu_tests.log
timeout -k 10 200 python scripts/driver_e2e.py --frames 2048 --batch 128 > $OUT/${TAG}_e2e_native.log 2>&1
echo "e2e native exit $?"; tail -1 $OUT/${TAG}_e2e_native.log > $OUT/${TAG}_e2e_native.json; cut -c1-700 $OUT/${TAG}_e2e_native.json
RAIN_NATIVE_IO=0 timeout -k 10 200 python scripts/driver_e2e.py --frames 1024 --batch 128 > $OUT/${TAG}_e2e_general.log 2>&1
echo "e2e general exit $?"; tail -1 $OUT/${TAG}_e2e_general.log > $OUT/${TAG}_e2e_general.json; cut -c1-700 $OUT/${TAG}_e2e_general.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp128
timeout -k 10 240 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/pp128 -- python $REPO/scripts/pipe_probe.py --copy-kernels 0 --pipe-batch 128 --rounds 10 --packed 1 > $OUT/${TAG}_pipe128.log 2>&1
python $REPO/scripts/pipe_timeline.py /tmp/pp128 > $OUT/${TAG}_timeline128.txt 2>&1
grep PIPE $OUT/${TAG}_pipe128.log | tail -2
grep "kernels " $OUT/${TAG}_timeline128.txt | cut -c1-110 | tail -8
