"""What the choice between the two restatements of cv2.fillConvexPoly changes (reference common/bad_weather.py:388; VERDICT r04
#5): the row-span rule of the fast colour kernels against OpenCV 3.2's own algorithm (oracle/cvlike.py cv_fill_convex_poly,
rr_device.h fov_rowspan_cv), on the KITTI 100 mm/hr frame of the benchmark scene, with the host build of the kernel arithmetic
(tests/hostemu: float64 throughout, no GPU needed).  python scripts/fill_rule_study.py [out.txt]"""
import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import helpers as h
from oracle import cvlike
from test_fov_f32_host import _polygons

lines = []
def say(s):
    print(s); lines.append(s)

tmp = pathlib.Path(tempfile.mkdtemp())
sc = h.Scene(tmp, 375, 1242, 8192, seed0=3000)
bg, env = sc.frame_inputs(0)
drops = sc.product_drops(0)
emu = h.hostemu()
outs = []
for rule in (0, 1):
    emu.emu_set_fill_rule(rule)
    outs.append(h.emu_render(sc, bg, bg, env, drops))
emu.emu_set_fill_rule(1)
a, b = outs
ok = a['status'] == 0
_, p64, n64, *_ = _polygons(sc, 0)
cvlike.set_fill_rule('cv', int(sc.cam.n_fov))
applies = np.array([n64[k] > 0 and cvlike.fill_rule_cv_applies(np.stack([p64[k, 0, :n64[k]], p64[k, 1, :n64[k]]], 1), sc.He, sc.We) for k in range(len(drops))])
px = []
for k in np.nonzero(applies)[0][:400]:
    P = np.stack([p64[k, 0, :n64[k]], p64[k, 1, :n64[k]]], 1)
    m1 = cvlike.fill_fov_mask_cv(np.zeros((sc.He, sc.We)), P)
    m0 = cvlike.fill_fov_mask_span_rule(np.zeros((sc.He, sc.We)), P)
    px.append((m0.sum(), m1.sum(), (m0 != m1).sum(), ((m0 == 1) & (m1 == 0)).sum()))
cvlike.set_fill_rule()
px = np.array(px)
say("KITTI 1242x375, map 1909x375, %d drops (%d composited); OpenCV's rule applies to %d polygons (%d wrap or fail and keep the span rule)"
    % (len(drops), int(ok.sum()), int(applies.sum()), int((~applies).sum())))
say("texels of a field of view (400 polygons): span rule %.0f, OpenCV's rule %.0f on average; %.0f differ (%.3f %%), of which %.1f are in the span rule only"
    % (px[:, 0].mean(), px[:, 1].mean(), px[:, 2].mean(), 100 * px[:, 2].mean() / px[:, 1].mean(), px[:, 3].mean()))
rel = np.abs(b['K'][ok] - a['K'][ok]) / np.abs(a['K'][ok])
say("colour constants K (3 per drop): max |dK| / |K| = %.3g, median %.3g, 99th percentile %.3g" % (rel.max(), np.median(rel), np.percentile(rel, 99)))
say("statuses equal: %s; rainy_mask (float64) equal: %s" % (np.array_equal(a['status'], b['status']), np.array_equal(a['mask'], b['mask'])))
df = np.abs(a['rainy_bg'] - b['rainy_bg'])
say("composite before the mean shift (float64): max |d| = %.3g = %.2f LSB of the uint8 image, 99.9th percentile %.3g" % (df.max(), df.max() * 255, np.percentile(df, 99.9)))
d = np.abs(a['image_u8'].astype(int) - b['image_u8'].astype(int))
touched = a['mask'] > 0
say("rainy_image uint8: max |d| = %d LSB; %.2f %% of all values differ (%.2f %% of the values of pixels some drop touches: %.0f %% of the frame); values off by >= 2 LSB: %d"
    % (d.max(), 100 * (d != 0).mean(), 100 * (d[touched] != 0).mean(), 100 * touched.mean(), int((d >= 2).sum())))
if len(sys.argv) > 1:
    open(sys.argv[1], 'w').write('\n'.join(lines) + '\n')
