#!/bin/bash
# One-GPU smoke of bench.py's N>1 path (run on the GPU box via gpurun): two ranks pinned to cuda:0, gloo instead
# of RCCL (RCCL refuses two ranks on one device), weak scaling and strong scaling (--total-frames) back to back.
# Checks the launch line the driver uses, the streak-DB broadcast, the barrier + max-over-ranks timing and that the
# two ranks of a strong-scaling run cover the sequence exactly once.  The numbers are NOT scaling results (one GPU).
# Usage: scripts/bench_two_ranks.sh [tag]   -> gpurun_out/two_ranks_<tag>.log
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO
export RAIN_BENCH_DEVICE=0 RAIN_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
COMMON="--gpus 2 --steps 2 --warmup 1 --batch 32 --no-cpu-baseline --no-prepass --no-variants --no-traffic"
{
  echo "== weak scaling, 2 ranks x 32 frames"
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py $COMMON
  echo "== strong scaling, one 64-frame sequence over 2 ranks"
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py $COMMON --total-frames 64
  echo "== same sequence on one rank"
  timeout 600 python bench.py --gpus 1 --steps 2 --warmup 1 --batch 32 --no-cpu-baseline --no-prepass --no-variants --no-traffic --total-frames 64
} > $OUT/two_ranks_$TAG.log 2>&1
grep -h '^{' $OUT/two_ranks_$TAG.log
