#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats + PMC passes of the bench command, compact per-kernel summary
# (text on stdout, JSON in gpurun_out/pmc_<tag>/summary.json: avg launch time, HBM-side traffic, VALU utilisation, LDS bank
# conflicts, L2 hit rate per kernel).
# Usage: scripts/gpu_pmc.sh <tag> "<bench args>" "<pass1 counters>" ["<pass2 counters>" ...]
# FETCH_SIZE and WRITE_SIZE each need a pass of their own (together: "exceeds the capabilities of the hardware", and
# rocprofv3 then sits on the aborted child until it is killed); every invocation runs under its own timeout.
TAG=$1; BARGS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (PMC_LEGS=1: not only the hot path -- the bench line's pre-pass / PNG / host-inclusive legs run too, so their kernels are profiled)
if [ "${PMC_LEGS:-0}" = "1" ]; then INNER=""; else INNER="--inner"; fi
BENCH="python $REPO/bench.py --steps 3 --warmup 1 $INNER $BARGS"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
i=0
for C in "$@"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pass$i -- $BENCH > $OUT/pass$i.log 2>&1 || echo "pass $i ($C) failed or timed out"
  i=$((i+1))
done
python - <<PY
import csv, glob, collections, re, json
out = "$OUT"
def short(k):
    m = re.search(r'(k_[a-z_0-9]+(?:<[^>]*>)?)', k)
    return m.group(1) if m else k[:40]
stats = {}
print('== kernel stats (avg us per launch)')
for f in glob.glob(out + '/stats/**/*kernel_stats.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        print('  %-28s calls %4s avg %10.1f us  %5s %%' % (short(row['Name']), row['Calls'], float(row['AverageNs']) / 1e3, row['Percentage']))
        stats[short(row['Name'])] = {"calls": int(row['Calls']), "avg_us": float(row['AverageNs']) / 1e3, "percent": float(row['Percentage'])}
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + '/pass*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = short(row.get('Kernel_Name', ''))
        agg[k][row['Counter_Name']] += float(row['Counter_Value']); cnt[(k, row['Counter_Name'])] += 1
print('== pmc (per-launch averages) and what follows from them')
summary = {}
for k in sorted(agg, key=lambda k: -stats.get(k, {}).get('avg_us', 0)):
    v = {c: x / cnt[(k, c)] for c, x in agg[k].items()}
    d = dict(stats.get(k, {}))
    if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:      # KB; FETCH_SIZE counts half of the bytes of wide reads on gfx950 (MI355X_MICROARCH.md)
        d['hbm_GB_per_launch'] = (2 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024 / 1e9
    if v.get('GRBM_GUI_ACTIVE') and 'SQ_INSTS_VALU' in v:      # GRBM_GUI_ACTIVE adds up the 8 XCDs; a wave64 VALU instruction holds its SIMD-32
        cyc = v['GRBM_GUI_ACTIVE'] / 8                          # for 4 cycles (float64) or 2 (float / integer): upper / lower bound
        d['valu_util_f64'] = 4 * v['SQ_INSTS_VALU'] / (cyc * 1024)
        d['valu_util_f32'] = 2 * v['SQ_INSTS_VALU'] / (cyc * 1024)
    if v.get('SQ_LDS_IDX_ACTIVE'):
        d['lds_bank_conflict_frac'] = v.get('SQ_LDS_BANK_CONFLICT', 0) / v['SQ_LDS_IDX_ACTIVE']
    if v.get('GRBM_GUI_ACTIVE') and 'SQ_LDS_IDX_ACTIVE' in v:
        d['lds_busy_frac'] = v['SQ_LDS_IDX_ACTIVE'] / (v['GRBM_GUI_ACTIVE'] / 8 * 256)      # LDS-array cycles per CU cycle
    if v.get('SQ_WAVE_CYCLES'):
        d['wait_any_frac'] = v.get('SQ_WAIT_ANY', 0) / v['SQ_WAVE_CYCLES']
        d['wait_inst_lds_frac'] = v.get('SQ_WAIT_INST_LDS', 0) / v['SQ_WAVE_CYCLES']
    if v.get('TCC_HIT_sum') is not None and (v.get('TCC_HIT_sum', 0) + v.get('TCC_MISS_sum', 0)) > 0:
        d['l2_hit'] = v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum'])
    d['counters'] = {c: round(x, 1) for c, x in v.items()}
    summary[k] = d
    print('  %-24s' % k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in d.items() if a != 'counters'})
json.dump(summary, open(out + '/summary.json', 'w'), indent=1)
PY
