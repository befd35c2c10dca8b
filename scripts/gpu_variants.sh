#!/bin/bash
# Run on the GPU box: rebuild librainhip.so with extra -D flags per variant and run the short bench for each.
# Usage: scripts/gpu_variants.sh <tag> "<flags variant 1>" "<flags variant 2>" ...     ("" = as committed)
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
OUT=$REPO/gpurun_out; mkdir -p $OUT
C=rain-rendering_amd/csrc
i=0
for FLAGS in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-value -Wno-unused-result \
     -Iinclude -I$C $FLAGS $C/rainhip.hip $C/rr_host.cpp $C/rr_png.cpp -lz -o $C/librainhip.so 2> $OUT/${TAG}_build$i.err || { echo "build failed: $FLAGS"; tail -5 $OUT/${TAG}_build$i.err; continue; }
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-prepass --no-variants --no-traffic 2> $OUT/${TAG}_v$i.err | tail -1 > $OUT/${TAG}_v$i.json
  python - <<PY
import json
try:
    d = json.load(open("$OUT/${TAG}_v$i.json"))
    print("[%s] %.0f frames/s  %.2f ms/step" % ("$FLAGS", d['value'], d['ms_per_step']))
    print("   ", {k: round(v, 2) for k, v in list(d['kernels_ms_per_call'].items())[:9]})
except Exception as e:
    print("[$FLAGS] failed:", e); print(open("$OUT/${TAG}_v$i.err").read()[-1500:])
PY
  i=$((i+1))
done
