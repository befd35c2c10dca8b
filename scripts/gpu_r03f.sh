#!/bin/bash
TAG=${1:-r03f}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for PB in 128 64; do
  rm -rf /tmp/pp$PB
  timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/pp$PB -- python $REPO/scripts/pipe_probe.py --copy-kernels 0 --pipe-batch $PB --rounds 8 --packed 1 > $OUT/${TAG}_pipe$PB.log 2>&1
  grep PIPE $OUT/${TAG}_pipe$PB.log
  python $REPO/scripts/pipe_timeline.py /tmp/pp$PB | cut -c1-220
done
