#!/bin/bash
TAG=${1:-r03e}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "pipeline or driver or abi or parity or particles_through or async" > $OUT/${TAG}_tests.log 2>&1; echo "tests exit $?"; tail -8 $OUT/${TAG}_tests.log
timeout -k 10 900 python bench.py --no-cpu-baseline --no-traffic > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench exit $?"
python - <<PY
import json
try:
    d = json.load(open("$OUT/${TAG}_bench.json"))
    print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 2))
    hi = d.get("host_inclusive", {})
    print("host_inclusive", round(hi.get("frames_per_s", 0)), hi.get("copies"), hi.get("pcie_GBps"))
    for k, v in (d.get("host_inclusive_variants") or {}).items():
        print("   ", k, round(v["frames_per_s"]), v["frames_per_slot"], v["copies"][:50], v["descriptors"])
except Exception as e:
    print("bench parse failed", e); print(open("$OUT/${TAG}_bench.err").read()[-3000:])
PY
cd /tmp && export TMPDIR=/tmp
python $REPO/scripts/pipe_probe.py --copy-kernels 0 --pipe-batch 128 --rounds 6 2>&1 | grep PIPE
