cd $GRAFT_REPO_ROOT
LEAN="--steps 5 --warmup 2 --no-cpu-baseline --no-prepass --no-variants --no-traffic"
for WL in "nuscenes1 64" "nuscenes5 64" "nuscenes25 64" "nuscenes100 64" "kitti25 128" "cityscapes50 32"; do set -- $WL
 timeout -k 10 400 python bench.py --workload $1 --batch $2 $LEAN > gpurun_out/r06g2_bench_$1.json 2> gpurun_out/r06g2_bench_$1.err
 python - <<PY
import json
d = json.load(open("gpurun_out/r06g2_bench_$1.json")); k=d['overlap']['kernels_ms_per_call_one_stream']
print("$1", round(d["value"], 1), round(d["ms_per_step"], 2), {x: round(k.get(x,0),2) for x in ('k_tile_rows','k_tile','k_tile_big')})
PY
done
timeout 600 python bench.py --steps 5 --warmup 2 --sweep "21=0" 2>&1 >/dev/null | grep SWEEP | cut -c1-400
