#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp128
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/pp128 -- python $REPO/scripts/pipe_probe.py --copy-kernels 0 --pipe-batch 128 --rounds 8 > $OUT/r03j_pipe128.log 2>&1
python $REPO/scripts/pipe_timeline.py /tmp/pp128 > $OUT/r03j_timeline128.txt
grep -v "void\b" $OUT/r03j_timeline128.txt | cut -c1-100 | head -50
