#!/bin/bash
# Quick GPU check (run via gpurun, ~5 min of box time): the GPU test tier, a lean headline bench and a lean in-kernel-particles bench.
# Usage: scripts/gpu_quick.sh <tag> [pytest -k expression]
TAG=${1:-quick}; KEXPR=${2:-}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO
if [ -n "$KEXPR" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $OUT/${TAG}_tests.log 2>&1
else
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_tests.log 2>&1
fi
echo "tests exit $?"; tail -15 $OUT/${TAG}_tests.log
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench exit $?"
python - <<PY
import json
try:
    d = json.load(open("$OUT/${TAG}_bench.json"))
    print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 2), "host_inclusive", round(d.get("host_inclusive", {}).get("frames_per_s", 0)))
    print({k: round(v, 2) for k, v in d["kernels_ms_per_call"].items()})
except Exception as e:
    print("bench parse failed", e); print(open("$OUT/${TAG}_bench.err").read()[-2000:])
PY
timeout -k 10 400 python bench.py --workload nuscenes100 --batch 32 --steps 3 --warmup 1 --no-traffic > $OUT/${TAG}_bench_nuscenes100.json 2> $OUT/${TAG}_bench_nuscenes100.err; echo "nuscenes bench exit $?"
python - <<PY
import json
try:
    d = json.load(open("$OUT/${TAG}_bench_nuscenes100.json"))
    print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 2)); print(d["config"]["workload"])
    print({k: round(v, 2) for k, v in d["kernels_ms_per_call"].items()})
except Exception as e:
    print("bench parse failed", e); print(open("$OUT/${TAG}_bench_nuscenes100.err").read()[-2000:])
PY
