#!/bin/bash
# Run on the GPU box (via gpurun): the host-inclusive pipeline under settings of the HIP runtime's copy path (the uploads are
# copy kernels of the runtime that share the compute units with the batch that is rendering: DESIGN section 6).
# Usage: scripts/gpu_copy_env.sh <tag> "<ENV=val ...>" ...      ("" = as is)
TAG=${1:-copyenv}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
: > $OUT/${TAG}.txt
for E in "$@"; do
  env $E timeout -k 10 300 python bench.py --batch 128 --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-traffic --no-driver > $OUT/${TAG}_tmp.json 2> $OUT/${TAG}_tmp.err
  python - <<PY | tee -a $OUT/${TAG}.txt
import json
try:
    d = json.load(open("$OUT/${TAG}_tmp.json"))
    h = d["host_inclusive"]
    print("%-50s host_inclusive %8.1f frames/s  (up %.1f GB/s, down %.1f GB/s)  value %.0f" % ("[$E]", h["frames_per_s"], h["pcie_GBps"]["up"], h["pcie_GBps"]["down"], d["value"]))
except Exception as e:
    print("[$E] failed:", e)
PY
done
