#!/bin/bash
# Run on the GPU box (via gpurun): GPU parity tests, then short bench runs (one JSON line each) for A/B variants.
# Usage: scripts/gpu_quick.sh <tag> [pytest -k expression]    -> gpurun_out/<tag>_{tests,bench*}.log
TAG=${1:-quick}; KEXPR=${2:-}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO
if [ -n "$KEXPR" ]; then
  timeout 1500 python -m pytest tests -q -m gpu -x -k "$KEXPR" 2>&1 | tail -15 > $OUT/${TAG}_tests.log
else
  timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $OUT/${TAG}_tests.log
fi
cat $OUT/${TAG}_tests.log
i=0
while read -r line; do
  [ -z "$line" ] && continue
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prepass --no-variants --no-traffic $line 2> $OUT/${TAG}_bench$i.err | tail -1 > $OUT/${TAG}_bench$i.json
  python - <<PY
import json
try:
    d = json.load(open("$OUT/${TAG}_bench$i.json"))
    print("bench[$line]: %.0f frames/s  %.2f ms/step" % (d['value'], d['ms_per_step']))
    print("   ", {k: round(v, 3) for k, v in d['kernels_ms_per_call'].items()})
except Exception as e:
    print("bench[$line] failed:", e); print(open("$OUT/${TAG}_bench$i.err").read()[-2000:])
PY
  i=$((i+1))
done < ${BENCH_VARIANTS:-/dev/null}
