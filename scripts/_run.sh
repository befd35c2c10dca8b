cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_properties.py -m gpu -q -x 2>&1 | tail -3
timeout -k 10 600 python bench.py --steps 5 --warmup 2 --opt 21=0 --sweep "22=0" --sweep "22=1" > gpurun_out/r06b_sweep.out 2> gpurun_out/r06b_sweep.err; grep '^SWEEP' gpurun_out/r06b_sweep.err | cut -c1-700
timeout -k 10 600 scripts/phase_timing.sh r06b --opt 21=0 > gpurun_out/r06b_phases.log 2>&1; grep -A8 '^PHASES' gpurun_out/r06b_phases.txt | cut -c1-900
