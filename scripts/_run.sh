cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
timeout -k 10 900 python bench.py --no-cpu-baseline --no-prepass --no-traffic > gpurun_out/r06n_bench.json 2> gpurun_out/r06n_bench.err; echo "bench exit $?"; python - <<'PY'
import json
d = json.load(open('gpurun_out/r06n_bench.json'))
print(d['value'], d['ms_per_step'])
print({k: (round(v.get('frames_per_s', 0)) if isinstance(v, dict) else v) for k, v in d.get('variants', {}).items()})
print({k: round(v, 2) for k, v in d['kernels_ms_per_call'].items()})
print({k: round(v, 2) for k, v in d.get('overlap', {}).get('kernels_ms_per_call_one_stream').items()})
print(d.get('host_inclusive_frames_per_s'), d['config'].get('raw_tile_share'), d.get('driver_end_to_end'))
PY
