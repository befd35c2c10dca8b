#!/bin/bash
TAG=${1:-r03i}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "pipeline or driver or abi or parity or particles_through or async or depth" > $OUT/${TAG}_tests.log 2>&1; echo "tests exit $?"; tail -5 $OUT/${TAG}_tests.log
cd /tmp && export TMPDIR=/tmp
for PB in 128 64; do
  rm -rf /tmp/pp$PB
  timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/pp$PB -- python $REPO/scripts/pipe_probe.py --copy-kernels 0 --pipe-batch $PB --rounds 8 > $OUT/${TAG}_pipe$PB.log 2>&1
  grep PIPE $OUT/${TAG}_pipe$PB.log
  python $REPO/scripts/pipe_timeline.py /tmp/pp$PB | cut -c1-200
done
for PB in 128 64 32 256; do python $REPO/scripts/pipe_probe.py --copy-kernels 0 --pipe-batch $PB --rounds 10 2>&1 | grep PIPE; done
