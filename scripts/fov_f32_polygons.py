"""Float32 field-of-view polygons against the float64 ones on a bench scene, on the host (tests/hostemu; glibc's atan2f
in place of the device's): how many drops fall back to float64, whether any status / vertex count differs, how far the
float vertices land from the float64 ones.  python scripts/fov_f32_polygons.py [--frames 8] [--cam KITTI|NUSCENES|CITYSCAPES]"""
import argparse, ctypes, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import helpers as th

ap = argparse.ArgumentParser()
ap.add_argument('--frames', type=int, default=8); ap.add_argument('--N', type=int, default=8192)
ap.add_argument('--H', type=int, default=375); ap.add_argument('--W', type=int, default=1242); ap.add_argument('--cam', default='KITTI')
a = ap.parse_args()
emu = th.hostemu()
PLAN = emu.emu_sizeof_plan()
sc = th.Scene(tempfile.mkdtemp(), a.H, a.W, a.N, n_frames=a.frames, cam=getattr(th, a.cam), seed0=3000)
texels, hs, ws, offs = th.hb.pack_streak_db(sc.db.streaks_light)
reasons = []; ratios = []; pxs = []; tot = unsure = bad_n = 0; moved = []; maxd = 0
for i in range(a.frames):
    drops = np.ascontiguousarray(sc.product_drops(i)); n = len(drops)
    plans = np.zeros(n * PLAN, np.uint8); p64 = np.zeros(n * 72, np.int32); n64 = np.zeros(n, np.int32); sizes = np.zeros(n, np.int64)
    emu.emu_plan(th._p(drops), n, ctypes.byref(sc.cam), a.H, a.W, sc.He, sc.We, th._p(hs), th._p(ws), ctypes.c_double(1.0), th._p(plans), th._p(p64), th._p(n64), th._p(sizes))
    p32 = np.zeros(n * 72, np.int32); n32 = np.zeros(n, np.int32); u32 = np.zeros(n, np.int32)
    emu.emu_fov_auto(th._p(drops), n, ctypes.byref(sc.cam), sc.He, sc.We, th._p(p32), th._p(n32), th._p(u32))
    ratio = np.zeros(n); mpx = np.zeros(n)
    emu.emu_fov_error_ratio(th._p(drops), n, ctypes.byref(sc.cam), sc.He, sc.We, th._p(ratio), th._p(mpx))
    ratios.append(ratio[ratio >= 0]); pxs.append(mpx[ratio >= 0])
    tot += n; unsure += int((u32 <= 0).sum()); reasons += [int(-v) for v in u32 if v <= 0]; bad_n += int((n32 != n64).sum())
    A, B = p32.reshape(n, 2, 36), p64.reshape(n, 2, 36)
    for k in range(n):
        if n32[k] == n64[k] and n32[k] > 0:
            d = np.abs(A[k, :, :n32[k]] - B[k, :, :n32[k]])
            maxd = max(maxd, int(d.max())); moved.append(int((d > 0).sum()))
print('drops', tot, 'float64 fall-backs', unsure, '(%.2f %%)' % (100.0 * unsure / tot), 'vertex-count mismatches', bad_n)
print('vertices moved per polygon: mean %.3f, max |delta| %d texel' % (np.mean(moved), maxd))
names = {1: 'setup', 2: 'discriminant', 4: 'pole', 8: 'seam', 16: 'wrap difference', 32: 'tiny polygon', 64: 'unsure or all sides alike'}
for b, nm in names.items():
    print('  %-28s %6d' % (nm, sum(1 for r in reasons if r & b)))
print('  only bit 64 + discriminant (beyond the radius):', sum(1 for r in reasons if (r & ~64) == 2))
r = np.concatenate(ratios); q = np.concatenate(pxs)
print('azimuth error / bound: max %.3f, p99.9 %.3f, median %.4f;  vertex offset in texels: max %.2e, median %.2e' % (r.max(), np.percentile(r, 99.9), np.median(r), q.max(), np.median(q)))
