#!/bin/bash
# The multi-GPU lines of BASELINE.json, each one command on an 8-GPU MI355X node (one process per GPU over RCCL; the only
# collective on the data path is the broadcast of the packed streak database).  Not run by the builder (no 8-GPU node in
# the build environment): the N>1 path is covered by the gloo world-2 tests and scripts/bench_two_ranks.sh.
#   weak scaling of the headline workload (what the driver's SCALE run does, N = 1, 2, 4, 8):
#     python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29500 bench.py --gpus N --steps 20 --warmup 5
#   configs[3]: ONE 1000-frame Cityscapes sequence (2048x1024, 50 mm/hr), frames idx[rank::8], strong scaling:
N=${1:-8}
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29501 bench.py --gpus $N \
  --workload cityscapes50 --total-frames 1000 --batch 25 --steps 3 --warmup 1
#   configs[4]: nuScenes 1600x900, the fall-rate sweep with in-kernel particles, weak scaling (each rank simulates its own frames):
for R in 1 5 25 100 200; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29502 bench.py --gpus $N \
    --workload nuscenes$R --batch 64 --steps 5 --warmup 2
done
#   the driver itself (main.py-compatible CLI), 8 ranks, each with its share of every sequence's frames; host threads per rank:
#   RAIN_IO_THREADS (default 2 x the CPU quota of the process; see DESIGN.md "host side of 8 ranks"):
#     python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 rain-rendering_amd/main.py --dataset kitti --intensity 25
