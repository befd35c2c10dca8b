#!/bin/bash
# Where the latency-bound tile / blur kernels spend their wave time: a build of the library with phase clocks (-DRR_PHASES,
# into /tmp, never the product's .so), the headline workload for a few steps, the clocks summed over all waves per phase.
# Usage (GPU box, via gpurun): scripts/phase_timing.sh <tag> [bench args]   -> gpurun_out/<tag>_phases.txt
TAG=${1:-phases}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$REPO/gpurun_out; mkdir -p $OUT /tmp/rr_phases
CS=$REPO/rain-rendering_amd/csrc
set -e
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-unused-result -DRR_PHASES -I$REPO/include -I$CS -c $CS/rainhip.hip -o /tmp/rr_phases/rainhip.o
for f in rr_host rr_png; do g++ -O2 -std=c++17 -ffp-contract=off -fPIC -pthread -I$REPO/include -I$CS -c $CS/$f.cpp -o /tmp/rr_phases/$f.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared /tmp/rr_phases/rainhip.o /tmp/rr_phases/rr_host.o /tmp/rr_phases/rr_png.o -lz -lpthread -o /tmp/rr_phases/librainhip.so
set +e
cd $REPO
RAINHIP_LIB=/tmp/rr_phases/librainhip.so timeout -k 10 400 python bench.py --steps 3 --warmup 1 --inner --phases "$@" 2> $OUT/${TAG}_phases.txt
grep -A40 "^PHASES" $OUT/${TAG}_phases.txt
