#!/bin/bash
# One GPU call that re-measures everything a round reports (run via gpurun, ~12 min of box time):
#   1. pytest -m gpu                          -> gpurun_out/<tag>_tests.log
#   2. python bench.py (default: the JSON line with roofline / cpu_baseline / variants / host-inclusive)
#                                             -> gpurun_out/<tag>_bench.json
#   3. rocprofv3 kernel stats + PMC passes (scripts/gpu_pmc.sh; FETCH_SIZE and WRITE_SIZE in passes of their own)
#                                             -> gpurun_out/pmc_<tag>/, gpurun_out/pmc_<tag>.txt
#   4. driver end to end (PNG in -> PNG out)  -> gpurun_out/<tag>_e2e.json
#   5. two ranks on the one GPU (gloo)        -> gpurun_out/two_ranks_<tag>.log
# Every step runs under its own timeout; a step that fails does not stop the next.
# Usage: scripts/gpu_full_measure.sh <tag>
TAG=${1:-full}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_tests.log 2>&1; echo "tests exit $?"; tail -3 $OUT/${TAG}_tests.log
timeout -k 10 600 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench exit $?"; tail -c 400 $OUT/${TAG}_bench.json
timeout -k 10 900 scripts/gpu_pmc.sh $TAG "" "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
  "TCC_HIT_sum TCC_MISS_sum" > $OUT/pmc_$TAG.txt 2>&1; echo "pmc exit $?"; head -25 $OUT/pmc_$TAG.txt
timeout -k 10 300 python scripts/driver_e2e.py --frames 512 2> $OUT/${TAG}_e2e.err | tail -1 > $OUT/${TAG}_e2e.json; echo "e2e exit $?"; cat $OUT/${TAG}_e2e.json
timeout -k 10 400 scripts/bench_two_ranks.sh $TAG; echo "two ranks exit $?"
