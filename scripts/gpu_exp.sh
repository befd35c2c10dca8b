#!/bin/bash
TAG=${1:-exp}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "parity or particles or edge or properties" > $OUT/${TAG}_tests.log 2>&1; echo "tests exit $?"; tail -4 $OUT/${TAG}_tests.log
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --sweep "1=1" > $OUT/${TAG}_sweep.out 2> $OUT/${TAG}_sweep.err; echo "sweep exit $?"; grep SWEEP $OUT/${TAG}_sweep.err | head -1 | cut -c1-520
for WL in "nuscenes100 64" "nuscenes200x 64"; do
  set -- $WL
  timeout -k 10 400 python bench.py --workload $1 --batch $2 --steps 5 --warmup 2 --no-cpu-baseline --no-prepass --no-variants --no-traffic > $OUT/${TAG}_bench_$1.json 2> $OUT/${TAG}_bench_$1.err
  python - <<PY
import json
d = json.load(open("$OUT/${TAG}_bench_$1.json"))
print("$1", round(d["value"], 1), "frames/s", round(d["ms_per_step"], 2), "ms;", {k: round(v, 2) for k, v in list(d["kernels_ms_per_call"].items())[:8]})
PY
done
