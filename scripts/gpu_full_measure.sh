#!/bin/bash
# Re-measures what a round reports, in two GPU calls (run via gpurun; every step under its own timeout, a failing step does
# not stop the next).  Results land in gpurun_out/; copy what is to be judged into profiles/.
#   scripts/gpu_full_measure.sh <tag> a     (~12 min of box time)
#     1. pytest -m gpu                                          -> <tag>_gpu_tests.log
#     2. python bench.py (the driver's line: roofline, valu, cpu_baseline, variants, host-inclusive pipeline, pre-pass)
#                                                               -> <tag>_bench_default.json
#     3. rocprofv3 kernel stats + PMC passes (scripts/gpu_pmc.sh: FETCH_SIZE and WRITE_SIZE in passes of their own,
#        VALU / wait / LDS / L2 counters)                       -> pmc_<tag>/summary.json, pmc_<tag>.txt
#     3b. phase clocks of the tile / blur kernels (scripts/phase_timing.sh: a -DRR_PHASES build in /tmp) -> <tag>_phases.txt
#   scripts/gpu_full_measure.sh <tag> b     (~12 min)
#     4. one lean bench line per other BASELINE configuration   -> <tag>_bench_<workload>.json
#     5. driver end to end (PNG in -> PNG out, main.py), PNG payloads from the device / deflated by the host
#                                                               -> <tag>_e2e.json, <tag>_e2e_host_deflate.json
#     6. two ranks on the one GPU (gloo): the N>1 launch line   -> two_ranks_<tag>.log
#     7. host CPU scaling probe (deflate threads)               -> <tag>_cpuscale.txt
TAG=${1:-full}; PART=${2:-a}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT $OUT/pmc_$TAG
cd $REPO
if [ "$PART" = "a" ]; then
  timeout -k 10 1200 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gpu_tests.log 2>&1; echo "tests exit $?"; tail -3 $OUT/${TAG}_gpu_tests.log
  timeout -k 10 900 python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err; echo "bench exit $?"; tail -c 300 $OUT/${TAG}_bench_default.json
  # kernel stats of the default step (colour branch on the second stream: its kernels' durations include waiting for room) ...
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pmc_$TAG/stats_overlap -- python $REPO/bench.py --steps 3 --warmup 1 --inner > $OUT/pmc_$TAG.overlap.log 2>&1); echo "stats (overlap) exit $?"
  # ... and stats + counters with everything on one stream (RR_OPT_COLOUR_STREAM 0): every kernel alone on the device
  timeout -k 10 1500 scripts/gpu_pmc.sh $TAG "--opt 21=0" "FETCH_SIZE" "WRITE_SIZE" \
    "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
    "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
    "TCC_HIT_sum TCC_MISS_sum" > $OUT/pmc_$TAG.txt 2>&1; echo "pmc exit $?"; grep -A12 "== pmc" $OUT/pmc_$TAG.txt | cut -c1-300
  timeout -k 10 600 scripts/phase_timing.sh $TAG > $OUT/${TAG}_phases.log 2>&1; echo "phases exit $?"; grep -A6 "^PHASES" $OUT/${TAG}_phases.log | cut -c1-400
else
  LEAN="--steps 5 --warmup 2 --no-cpu-baseline --no-prepass --no-variants --no-traffic"
  for WL in "kitti25 128" "cityscapes50 32" "cityscapes50_rs2 128" "nuscenes1 64" "nuscenes5 64" "nuscenes25 64" "nuscenes100 64" "nuscenes200 64" "nuscenes200x 64"; do
    set -- $WL
    timeout -k 10 400 python bench.py --workload $1 --batch $2 $LEAN > $OUT/${TAG}_bench_$1.json 2> $OUT/${TAG}_bench_$1.err; echo "$1 exit $?"
    python - <<PY
import json
try:
    d = json.load(open("$OUT/${TAG}_bench_$1.json"))
    print("  ", round(d["value"], 1), "frames/s,", round(d["ms_per_step"], 2), "ms/step of", d["config"]["frames_per_call"], "frames; dominant", d["roofline"]["kernel"], round(d["roofline"]["avg_launch_ms"], 2), "ms, frac", round(d["roofline"]["frac"], 3))
except Exception as e:
    print("   parse failed", e)
PY
  done
  timeout -k 10 400 python scripts/driver_e2e.py --frames 2048 --batch 128 2> $OUT/${TAG}_e2e.err | tail -1 > $OUT/${TAG}_e2e.json; echo "e2e exit $?"; cut -c1-600 $OUT/${TAG}_e2e.json
  RAIN_PNG_DEVICE=0 timeout -k 10 400 python scripts/driver_e2e.py --frames 2048 --batch 128 2> $OUT/${TAG}_e2e_host_deflate.err | tail -1 > $OUT/${TAG}_e2e_host_deflate.json; echo "e2e (host deflate) exit $?"; cut -c1-600 $OUT/${TAG}_e2e_host_deflate.json
  timeout -k 10 500 scripts/bench_two_ranks.sh $TAG; echo "two ranks exit $?"
  timeout -k 10 120 python scripts/cpuscale_probe.py > $OUT/${TAG}_cpuscale.txt 2>&1; cat $OUT/${TAG}_cpuscale.txt
fi
