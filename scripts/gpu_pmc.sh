#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats + PMC passes of the bench command, compact per-kernel summary.
# Usage: scripts/gpu_pmc.sh <tag> "<bench args>" "<pass1 counters>" ["<pass2 counters>" ...]
# FETCH_SIZE and WRITE_SIZE each need a pass of their own (together: "exceeds the capabilities of the hardware", and
# rocprofv3 then sits on the aborted child until it is killed); every invocation runs under its own timeout.
TAG=$1; BARGS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 3 --warmup 1 --inner $BARGS"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
i=0
for C in "$@"; do
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pass$i -- $BENCH > $OUT/pass$i.log 2>&1 || echo "pass $i ($C) failed or timed out"
  i=$((i+1))
done
python - <<PY
import csv, glob, collections, re
out = "$OUT"
def short(k):
    m = re.search(r'(k_[a-z_0-9]+(?:<[^>]*>)?)', k)
    return m.group(1) if m else k[:40]
print('== kernel stats (avg us per launch)')
for f in glob.glob(out + '/stats/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        print('  %-28s calls %4s avg %10.1f us  %5s %%' % (short(row['Name']), row['Calls'], float(row['AverageNs']) / 1e3, row['Percentage']))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + '/pass*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row.get('Kernel_Name', ''))
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); cnt[(k, row['Counter_Name'])] += 1
print('== pmc (per-launch averages)')
for k in agg:
    print('  %-28s' % k, {c: round(v / cnt[(k, c)], 1) for c, v in agg[k].items()})
PY
