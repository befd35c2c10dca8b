cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_properties.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -3
timeout -k 10 600 python bench.py --steps 5 --warmup 2 --opt 21=0 --sweep "23=1" --sweep "23=2" --sweep "23=3" --sweep "23=4"  --sweep "23=6" > gpurun_out/r06g_sweep.out 2> gpurun_out/r06g_sweep.err; grep '^SWEEP' gpurun_out/r06g_sweep.err | cut -c1-700
timeout -k 10 600 scripts/phase_timing.sh r06g --opt 21=0 > gpurun_out/r06g_phases.log 2>&1; grep -A8 '^PHASES' gpurun_out/r06g_phases.txt | grep 'k_tile' | cut -c1-1100
