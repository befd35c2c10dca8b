#!/bin/bash
# One rocprofv3 --pmc pass of the bench command (run on the GPU box via gpurun).
# Usage: scripts/gpu_pmc_one.sh <tag> "<COUNTERS>" [bench args...]
set -u
TAG=$1; C=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-prepass "$@" > $OUT/log.txt 2>&1
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT" + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get('Kernel_Name', '')[:60]
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); cnt[(k, row['Counter_Name'])] += 1
for k in agg:
    print(k, {c: (v / cnt[(k, c)]) for c, v in agg[k].items()})
PY
