#!/bin/bash
TAG=${1:-r03d}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CK in 1 0; do
  rm -rf /tmp/pp$CK
  timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/pp$CK -- python $REPO/scripts/pipe_probe.py --copy-kernels $CK > $OUT/${TAG}_pipe$CK.log 2>&1
  grep PIPE $OUT/${TAG}_pipe$CK.log
  python $REPO/scripts/pipe_timeline.py /tmp/pp$CK | cut -c1-600
done
python $REPO/scripts/pipe_probe.py --copy-kernels 1 --pipe-batch 128 --rounds 6 2>&1 | grep PIPE
