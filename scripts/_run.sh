cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout -k 10 900 python bench.py --no-cpu-baseline --no-prepass --no-traffic > gpurun_out/r06h_bench.json 2> gpurun_out/r06h_bench.err; echo "bench exit $?"; python - <<'PY'
import json
d = json.load(open('gpurun_out/r06h_bench.json'))
print(d['value'], d['ms_per_step'], d.get('variants'))
print({k: round(v, 2) for k, v in d['kernels_ms_per_call'].items()})
print(d.get('overlap', {}).get('kernels_ms_per_call_one_stream'))
PY
