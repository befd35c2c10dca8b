"""Per-drop statistics of a bench workload from the host build of plan_drop (no GPU): what the tile / blur / compositor
kernels have to do per frame.  python scripts/workload_stats.py [--workload kitti100] [--frames 4]"""
import argparse, ctypes, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import helpers as th
hb = th.hb

PLAN = np.dtype([('status', 'i4'), ('kind', 'i4'), ('tex', 'i4'), ('flip', 'i4'), ('tw', 'i4'), ('th', 'i4'), ('shift', 'i4'), ('pw', 'i4'), ('ph', 'i4'),
                 ('r1', 'i4'), ('r2', 'i4'), ('vis_x0', 'i4'), ('vis_y0', 'i4'), ('vis_w', 'i4'), ('vis_h', 'i4'), ('crop_x', 'i4'), ('crop_y', 'i4'),
                 ('ew', 'i4'), ('bw0', 'i4'), ('nW', 'i4'), ('nH', 'i4'), ('rs_mode', 'i4'), ('isx', 'i4'), ('isy', 'i4'), ('eh', 'i4'), ('epitch', 'i4'),
                 ('epad', 'i4'), ('pad_', 'i4'), ('a0', 'i8'), ('a1', 'i8'), ('sig1', 'f8'), ('sig2', 'f8'), ('tau', 'f8'), ('g', 'f8'), ('mi', 'f8', 9),
                 ('ma', 'f8', 6), ('sx', 'f8'), ('sy', 'f8'), ('isx_', 'f8'), ('isy_', 'f8')])

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=4)
    ap.add_argument('--H', type=int, default=375); ap.add_argument('--W', type=int, default=1242); ap.add_argument('--N', type=int, default=8192)
    a = ap.parse_args()
    emu = th.hostemu()
    assert emu.emu_sizeof_plan() == PLAN.itemsize, (emu.emu_sizeof_plan(), PLAN.itemsize)
    sc = th.Scene(tempfile.mkdtemp(), a.H, a.W, a.N, n_frames=a.frames, cam=th.KITTI, seed0=3000)
    texels, hs, ws, offs = hb.pack_streak_db(sc.db.streaks_light)
    print('textures', len(hs), 'h', sorted(set(hs.tolist())), 'w', sorted(set(ws.tolist())))
    allp = []
    for i in range(a.frames):
        drops = np.ascontiguousarray(sc.product_drops(i)); n = len(drops)
        plans = np.zeros(n, PLAN); poly = np.zeros(n * 72, np.int32); npts = np.zeros(n, np.int32); sizes = np.zeros(n, np.int64)
        emu.emu_plan(th._p(drops), n, ctypes.byref(sc.cam), a.H, a.W, sc.He, sc.We, th._p(hs), th._p(ws), ctypes.c_double(1.0), th._p(plans), th._p(poly), th._p(npts), th._p(sizes))
        ok = (plans['status'] == 0) & (sizes > 0)
        allp.append(plans[ok])
        if i == 0:
            print('frame 0: drops', n, 'ok', ok.sum(), 'npts hist', np.unique(npts, return_counts=True))
            ys = poly.reshape(n, 2, 36)[:, 1, :]
            pp = [(ys[k, :npts[k]].min(), ys[k, :npts[k]].max()) for k in range(n) if npts[k] > 0]
            pp = np.array(pp); print('  polygon rows covered mean', (np.clip(pp[:, 1], 0, sc.He - 1) - np.clip(pp[:, 0], 0, sc.He - 1) + 1).mean(), 'of', sc.He)
    p = np.concatenate(allp); n = len(p) / a.frames
    def pr(name, v): print('  %-34s mean %9.1f  p50 %7.0f  p90 %7.0f  p99 %7.0f  max %7.0f   sum/frame %12.0f' % (name, v.mean(), np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.max(), v.sum() / a.frames))
    print('per frame ok drops', n)
    for kind, nm in ((0, 'BIG'), (1, 'ROT')):
        q = p[p['kind'] == kind]
        print(nm, 'per frame', len(q) / a.frames)
        pr('tw', q['tw']); pr('th', q['th']); pr('tw*th', q['tw'] * q['th'])
        if kind == 1:
            pr('nW', q['nW']); pr('nH', q['nH']); pr('nW*nH canvas', q['nW'] * q['nH']); pr('tex pixels', hs[q['tex']] * ws[q['tex']])
            print('  rs_mode', np.unique(q['rs_mode'], return_counts=True))
    pr('r1', p['r1']); pr('r2', p['r2']); pr('ew*eh', p['ew'] * p['eh']); pr('raw tw*th', p['tw'] * p['th'])
    bl = p[p['r1'] > 0]
    print('blurred per frame', len(bl) / a.frames)
    pr('row-pass MACs tw*eh*(r1+1)', bl['tw'] * bl['eh'] * (bl['r1'] + 1)); pr('col-pass MACs ew*eh*(r2+1)', bl['ew'] * bl['eh'] * (bl['r2'] + 1))
    pr('vis_w*vis_h (reference footprint)', p['vis_w'] * p['vis_h'])
    # distinct raw tiles across the frames
    key = np.stack([p['kind'], p['tex'], p['flip'], p['tw'], p['th']] + [p['mi'][:, k].view('i8') for k in range(9)] + [p['ma'][:, k].view('i8') for k in range(6)], 1)
    print('distinct raw tiles', len(np.unique(key, axis=0)), 'of', len(p))

if '--blur' not in sys.argv:
    main()


def blur_classes():
    """Which blur kernel takes which drops, and how the filter work splits by radius (same scene as above)."""
    import ctypes
    emu = th.hostemu()
    a_frames = 3
    sc = th.Scene(tempfile.mkdtemp(), 375, 1242, 8192, n_frames=a_frames, cam=th.KITTI, seed0=3000)
    texels, hs, ws, offs = hb.pack_streak_db(sc.db.streaks_light)
    rows = []
    for i in range(a_frames):
        drops = np.ascontiguousarray(sc.product_drops(i)); n = len(drops)
        plans = np.zeros(n, PLAN); poly = np.zeros(n * 72, np.int32); npts = np.zeros(n, np.int32); sizes = np.zeros(n, np.int64)
        emu.emu_plan(th._p(drops), n, ctypes.byref(sc.cam), 375, 1242, sc.He, sc.We, th._p(hs), th._p(ws), ctypes.c_double(1.0), th._p(plans), th._p(poly), th._p(npts), th._p(sizes))
        p = plans[(plans['status'] == 0) & (sizes > 0) & (plans['r1'] > 0)]
        out = np.zeros(4, np.int32)
        for q in p:
            emu.emu_blur_layout(int(q['ew']), int(q['eh']), int(q['r1']), int(q['r2']), int(q['tw']), int(q['th']), 2816, 2048, th._p(out))
            macs = int(q['tw']) * int(q['eh']) * (int(q['r1']) + 1) + int(q['ew']) * int(q['eh']) * (int(q['r2']) + 1)
            rows.append(('small' if out[0] else ('fused' if out[1] else 'slow'), int(q['r1']), macs, int(q['ew']) * int(q['eh'])))
    for cls in ('small', 'fused', 'slow'):
        r = [x for x in rows if x[0] == cls]
        if not r:
            continue
        r1 = np.array([x[1] for x in r]); m = np.array([x[2] for x in r]); px = np.array([x[3] for x in r])
        print('%s: %d drops/frame, MACs/frame %.3g (mean %d), output px/frame %.3g; r1 mean %.1f' % (cls, len(r) / a_frames, m.sum() / a_frames, m.mean(), px.sum() / a_frames, r1.mean()))
        for lo, hi in ((1, 1), (2, 2), (3, 4), (5, 8), (9, 16), (17, 48)):
            k = (r1 >= lo) & (r1 <= hi)
            print('   r1 %2d..%2d: %5.1f %% of drops, %5.1f %% of MACs' % (lo, hi, 100.0 * k.mean(), 100.0 * m[k].sum() / m.sum()))


if '--blur' in sys.argv:
    blur_classes()
