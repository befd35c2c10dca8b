#!/bin/bash
# Run on the GPU box (via gpurun): the pre-pass / pipeline GPU tests, then one bench line with the pre-pass and host-inclusive legs.
# Usage: scripts/gpu_prepass_check.sh <tag> [bench args]
TAG=${1:-pre}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
timeout -k 10 900 python -m pytest tests/test_gpu_prepass.py tests/test_gpu_pipeline_async.py tests/test_gpu_driver.py -m gpu -q > $OUT/${TAG}_tests.log 2>&1; echo "tests exit $?"; tail -12 $OUT/${TAG}_tests.log
timeout -k 10 900 python bench.py --batch 256 --steps 5 --warmup 2 --no-cpu-baseline --no-variants --no-traffic --no-driver "$@" > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench exit $?"
python - <<PY
import json
try:
    d = json.load(open("$OUT/${TAG}_bench.json"))
    print(round(d["value"], 1), "frames/s", round(d["ms_per_step"], 2), "ms/step")
    for k in ("prepass", "png_on_device", "host_inclusive"):
        print(k, json.dumps(d.get(k))[:1400])
except Exception as e:
    print("parse failed", e)
PY
tail -5 $OUT/${TAG}_bench.err
