"""How long hipMalloc / hipHostMalloc take as a function of size (the driver's set-up clock: the first rr_pipeline_submit of a
run allocates ~8 GB of scratch for 128 KITTI frames).  python scripts/probes/malloc_cost.py   (GPU box)"""
import ctypes, time
hip = ctypes.CDLL('/opt/rocm/lib/libamdhip64.so')
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
hip.hipFree.argtypes = [ctypes.c_void_p]
hip.hipHostFree.argtypes = [ctypes.c_void_p]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
t0 = time.time(); hip.hipSetDevice(0); p = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(p), 1 << 20); hip.hipDeviceSynchronize(); print('context + first MB: %.3f s' % (time.time() - t0))
for rep in range(2):
    for mb in (16, 256, 1024, 4096):
        t0 = time.time(); q = ctypes.c_void_p(); rc = hip.hipMalloc(ctypes.byref(q), mb << 20); t1 = time.time()
        hip.hipMemset(q, 0, mb << 20); hip.hipDeviceSynchronize(); t2 = time.time()
        hip.hipFree(q); t3 = time.time()
        print('hipMalloc %5d MB: %.1f ms (rc %d), memset %.1f ms, free %.1f ms' % (mb, 1e3 * (t1 - t0), rc, 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
t0 = time.time()
ps = []
for k in range(50):
    q = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(q), 160 << 20); ps.append(q)
print('50 x 160 MB: %.1f ms' % (1e3 * (time.time() - t0)))
for q in ps: hip.hipFree(q)
for mb in (64, 1024):
    t0 = time.time(); q = ctypes.c_void_p(); rc = hip.hipHostMalloc(ctypes.byref(q), mb << 20, 0); t1 = time.time(); hip.hipHostFree(q)
    print('hipHostMalloc %5d MB: %.1f ms (rc %d)' % (mb, 1e3 * (t1 - t0), rc))
