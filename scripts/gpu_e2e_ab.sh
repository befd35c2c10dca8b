#!/bin/bash
# Run on the GPU box (via gpurun): the driver end to end with the device's and with the host's deflate, alternating (the box's
# host side is noisy: one pair says little).  Usage: scripts/gpu_e2e_ab.sh <tag> [rounds] [switch]
# switch: the environment variable to alternate between 1 and 0 (default RAIN_PNG_DEVICE: the output files' deflate on the
# device / on the host; RAIN_PNG_ROWS: the input files' scanline filters reversed on the device / by the host).
TAG=${1:-e2e}; ROUNDS=${2:-3}; SW=${3:-RAIN_PNG_DEVICE}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
: > $OUT/${TAG}_e2e_ab.jsonl
for R in $(seq 1 $ROUNDS); do
  for DEV in 1 0; do
    env $SW=$DEV timeout -k 10 500 python scripts/driver_e2e.py --frames 2048 --batch 128 2> $OUT/${TAG}_e2e_dev$DEV.err | tail -1 > $OUT/${TAG}_tmp.json
    python - <<PY
import json
d = json.load(open("$OUT/${TAG}_tmp.json"))
t = d["timing"][0]
line = {"$SW": $DEV, "round": $R, "steady_frames_per_s": t["steady_frames_per_s"], "frames_per_s_including_setup": d["frames_per_s"], "first_batch_s": t["first_batch_s"], "frames": d["frames"], "cpu_quota": d["cpu_quota"]}
print(json.dumps(line))
open("$OUT/${TAG}_e2e_ab.jsonl", "a").write(json.dumps(line) + "\n")
PY
  done
done
