#!/bin/bash
# Texture staging from pre-padded copies (RR_OPT_PADDED_TEXTURES, default on): A/B against the byte-wise staging in one
# process, then the whole GPU test tier, the default bench line and the driver end to end on the new default.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
TAG=${1:-r03l}
cd $REPO
timeout -k 10 150 python bench.py --steps 6 --warmup 2 --sweep "9=0" > $OUT/${TAG}_sweep.out 2> $OUT/${TAG}_sweep.err; echo "sweep exit $?"; grep SWEEP $OUT/${TAG}_sweep.err | cut -c1-420
timeout -k 10 330 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/${TAG}_gpu_tests.log 2>&1
echo "tests exit $?"; tail -3 $OUT/${TAG}_gpu_tests.log
timeout -k 10 60 python scripts/driver_e2e.py --frames 2048 --batch 128 > $OUT/${TAG}_e2e_native.log 2>&1
echo "e2e native exit $?"; tail -1 $OUT/${TAG}_e2e_native.log > $OUT/${TAG}_e2e_native.json; cut -c1-600 $OUT/${TAG}_e2e_native.json
timeout -k 10 260 python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
echo "bench exit $?"; python - <<PY
import json
try:
    d = json.load(open('$OUT/${TAG}_bench_default.json'))
    print(round(d['value']), d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], {k: round(v, 2) for k, v in list(d['kernels_ms_per_call'].items())[:6]}, round(d['host_inclusive']['frames_per_s']))
except Exception as e:
    print('no bench line', e)
PY
