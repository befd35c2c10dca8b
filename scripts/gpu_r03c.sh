#!/bin/bash
TAG=${1:-r03c}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "pipeline or driver or abi or parity" > $OUT/${TAG}_tests.log 2>&1; echo "tests exit $?"; tail -8 $OUT/${TAG}_tests.log
timeout -k 10 600 python bench.py --no-cpu-baseline --no-traffic > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench exit $?"
python - <<PY
import json
try:
    d = json.load(open("$OUT/${TAG}_bench.json"))
    print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 2))
    print("host_inclusive", json.dumps(d.get("host_inclusive"))[:600])
    print("variants", json.dumps(d.get("host_inclusive_variants"))[:1200])
    print({k: round(v, 2) for k, v in d["kernels_ms_per_call"].items()})
except Exception as e:
    print("bench parse failed", e); print(open("$OUT/${TAG}_bench.err").read()[-3000:])
PY
timeout -k 10 1500 scripts/gpu_pmc.sh $TAG "" "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "TCC_HIT_sum TCC_MISS_sum" > $OUT/pmc_$TAG.txt 2>&1; echo "pmc exit $?"; cat $OUT/pmc_$TAG.txt | cut -c1-330
