#!/bin/bash
# A/B builds of the library with extra -D flags: scripts/build_alt.sh <tag> [-DNAME=value ...]  -> build_alt/librainhip_<tag>.so
# (git-ignored, travels with gpurun; select it with RAINHIP_LIB=build_alt/librainhip_<tag>.so)
TAG=$1; shift
REPO=$(cd $(dirname $0)/.. && pwd); CS=$REPO/rain-rendering_amd/csrc; mkdir -p $REPO/build_alt
set -e
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -Wno-unused-result "$@" -I$REPO/include -I$CS -c $CS/rainhip.hip -o $REPO/build_alt/rainhip_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $REPO/build_alt/rainhip_$TAG.o $CS/rr_host.o $CS/rr_png.o -lz -lpthread -o $REPO/build_alt/librainhip_$TAG.so
rm -f $REPO/build_alt/rainhip_$TAG.o
