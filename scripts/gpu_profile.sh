#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats and PMC passes of the bench command.
# Usage: scripts/gpu_profile.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/{stats,pmc_*}/...  Copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BATCH=${RAIN_PROFILE_BATCH:-256}
BENCH="python $REPO/bench.py --steps 5 --warmup 2 --batch $BATCH --no-cpu-baseline --no-prepass $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
# RAIN_PROFILE_QUICK=1: only the two HBM-traffic passes (each pass re-runs the bench incl. its scene set-up)
if [ "${RAIN_PROFILE_QUICK:-0}" = "2" ]; then
  PASSES=()                        # kernel-trace statistics only
elif [ "${RAIN_PROFILE_QUICK:-0}" = "1" ]; then
  PASSES=("FETCH_SIZE" "WRITE_SIZE")
else
  PASSES=("FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum")
fi
for C in "${PASSES[@]}"; do
  NAME=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$NAME -- $BENCH > $OUT/pmc_$NAME.log 2>&1
done
# compact summaries
python - <<PY
import csv, glob, os, collections
out = "$OUT"
for f in glob.glob(out + '/stats/**/*kernel_stats.csv', recursive=True):
    print('== kernel stats', f)
    print(open(f).read())
import json
traffic = {}
for d in sorted(glob.glob(out + '/pmc_*')):
    if not os.path.isdir(d): continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get('Kernel_Name', '')
            agg[k][row['Counter_Name']] += float(row['Counter_Value'])
            cnt[(k, row['Counter_Name'])] += 1
    print('== pmc', os.path.basename(d))
    for k in agg:
        print('  ', k[:70], {c: (v, cnt[(k, c)]) for c, v in agg[k].items()})
        for c in ('FETCH_SIZE', 'WRITE_SIZE'):
            if c in agg[k]:
                import re
                m = re.search(r'(k_[a-z_0-9]+(?:<\d>)?)', k)  # k_blur<0>, k_blur<1> stay distinct
                if m and re.match(r'k_colour_bands', m.group(1)): m = re.search(r'(k_colour_bands)', k)
                name = m.group(1) if m else k[:40]
                traffic.setdefault(name, {})[c + '_KB_per_launch'] = agg[k][c] / cnt[(k, c)]
# MI355X_MICROARCH.md (HBM): bytes = KB * 1024; on gfx950 FETCH_SIZE under-reports wide reads by 2x -> doubled
for name, t in traffic.items():
    if 'FETCH_SIZE_KB_per_launch' in t and 'WRITE_SIZE_KB_per_launch' in t:
        t['hbm_bytes_per_launch'] = (2.0 * t['FETCH_SIZE_KB_per_launch'] + t['WRITE_SIZE_KB_per_launch']) * 1024.0
json.dump({'batch': int("$BATCH"), 'workload': [1242, 375, 100], 'correction': '2*FETCH_SIZE + WRITE_SIZE, KB*1024 (MI355X_MICROARCH.md HBM section)',
           'kernels': traffic}, open(out + '/traffic.json', 'w'), indent=1)
PY
