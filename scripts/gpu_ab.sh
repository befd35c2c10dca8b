#!/bin/bash
# Run on the GPU box (via gpurun): a GPU parity subset, then ONE bench process that times the headline configuration and a
# list of rr_set_option sets (bench.py --sweep, scene set-up paid once), then optional extra bench invocations.
# Usage: scripts/gpu_ab.sh <tag> "<pytest files / -k expression or empty = the whole GPU tier>" "<sweep 1>" ... [-- "<bench args>" ...]
TAG=${1:-ab}; SEL=${2:-}; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT; cd $REPO
timeout -k 10 1500 python -m pytest ${SEL:-tests} -m gpu -q > $OUT/${TAG}_tests.log 2>&1; echo "tests exit $?"; tail -6 $OUT/${TAG}_tests.log
ARGS=""
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS="$ARGS --sweep $1"; shift; done
timeout -k 10 600 python bench.py --steps 5 --warmup 2 --sweep "1=1" $ARGS > $OUT/${TAG}_sweep.out 2> $OUT/${TAG}_sweep.err; echo "sweep exit $?"; grep '^SWEEP' $OUT/${TAG}_sweep.err | cut -c1-620
if [ "$1" = "--" ]; then shift; fi
i=0
for B in "$@"; do
  timeout -k 10 600 python bench.py $B > $OUT/${TAG}_bench$i.json 2> $OUT/${TAG}_bench$i.err; echo "bench [$B] exit $?"
  python - <<PY
import json
try:
    d = json.load(open("$OUT/${TAG}_bench$i.json"))
    print("  ", round(d["value"], 1), "frames/s,", round(d["ms_per_step"], 2), "ms/step of", d["config"]["frames_per_call"], "frames;", {k: round(v, 2) for k, v in list(d["kernels_ms_per_call"].items())[:9]})
except Exception as e:
    print("   parse failed", e)
PY
  i=$((i+1))
done
