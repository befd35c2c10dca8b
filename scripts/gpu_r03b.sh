#!/bin/bash
TAG=${1:-r03b}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_tests.log 2>&1; echo "tests exit $?"; tail -15 $OUT/${TAG}_tests.log
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --sweep "7=1" > $OUT/${TAG}_sweep.out 2> $OUT/${TAG}_sweep.err; echo "sweep exit $?"; grep SWEEP $OUT/${TAG}_sweep.err
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
hipcc --offload-arch=gfx950 -O3 scripts/probes/pcie_probe.hip -o /tmp/pcie_probe 2>/dev/null && timeout 120 /tmp/pcie_probe > $OUT/${TAG}_pcie_probe.txt 2>&1; cat $OUT/${TAG}_pcie_probe.txt
