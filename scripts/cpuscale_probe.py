import os, time, zlib, numpy as np
from concurrent.futures import ThreadPoolExecutor
print('cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'n/a', 'nproc', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
rows = np.random.RandomState(0).randint(0, 255, 375*4969, dtype=np.uint8)
rows = (np.cumsum(np.random.RandomState(0).randint(-2, 3, 375*4969)) % 256).astype(np.uint8).tobytes()
t0=time.time(); zlib.compress(rows, 1); print('single %.1f ms'%((time.time()-t0)*1e3))
for T in (8, 16, 32, 64, 128):
    with ThreadPoolExecutor(T) as ex:
        t0=time.time(); list(ex.map(lambda _: zlib.compress(rows, 1), range(4*T))); dt=time.time()-t0
        print('T=%d: %.0f compress/s'%(T, 4*T/dt))
