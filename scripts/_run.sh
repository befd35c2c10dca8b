cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_properties.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -4
timeout -k 10 600 python bench.py --steps 5 --warmup 2 --opt 21=0 --sweep "18=0" --sweep "18=1" > gpurun_out/r06m_sweep.out 2> gpurun_out/r06m_sweep.err; grep '^SWEEP' gpurun_out/r06m_sweep.err | cut -c1-420
timeout -k 10 600 python bench.py --steps 5 --warmup 2 --sweep "18=0" --sweep "18=1" > gpurun_out/r06m_sweep2.out 2> gpurun_out/r06m_sweep2.err; grep '^SWEEP' gpurun_out/r06m_sweep2.err | cut -c1-300
