"""CPU tier: the two restatements of cv2.fillConvexPoly(mask, s, 1) (reference common/bad_weather.py:388; OpenCV is not
installed here: both UNPINNED) -- the row-span rule of the fast colour kernels and OpenCV 3.2's own algorithm (Bresenham
outline + 16.16 edge walkers), which oracle/cvlike.py restates literally (cv_fill_convex_poly) and rr_device.h in closed
form per edge and row (fov_rowspan_cv; the library's RR_OPT_FOV_FILL_RULE 1).  Checked here: answers that follow from the
published algorithm, the closed forms (C++ and numpy) against the literal restatement on the polygons of test scenes in
every rotation and orientation of the vertex list (Clipper's starting vertex is unknown: for these polygons it cannot
matter), and how far apart the two rules are -- colour only, the mask never sees the polygon."""
import ctypes

import numpy as np

import helpers as h
from oracle import cvlike
from test_fov_f32_host import _polygons


def _spans_of_mask(m):
    xl, xr = np.ones(m.shape[0], np.int64), np.zeros(m.shape[0], np.int64)
    for y in range(m.shape[0]):
        nz = np.nonzero(m[y])[0]
        if len(nz):
            assert nz[-1] - nz[0] + 1 == len(nz), "row %d is not one run" % y
            xl[y], xr[y] = nz[0], nz[-1]
    return xl, xr


def test_known_answers_of_the_opencv_fill():
    # an axis-parallel rectangle: both borders included
    m = cvlike.cv_fill_convex_poly(np.zeros((12, 16)), [(2, 3), (10, 3), (10, 8), (2, 8)])
    assert m.sum() == 9 * 6 and m[3:9, 2:11].all()
    # Line(): the 8-connected Bresenham line from the left end point; a slope of 1/2 steps down on every second pixel
    assert cvlike.cv_line_pixels(20, 20, (0, 0), (6, 3)) == [(0, 0), (1, 0), (2, 1), (3, 1), (4, 2), (5, 2), (6, 3)]
    assert cvlike.cv_line_pixels(20, 20, (6, 3), (0, 0)) == cvlike.cv_line_pixels(20, 20, (0, 0), (6, 3))
    assert cvlike.cv_line_pixels(20, 20, (3, 1), (3, 5)) == [(3, y) for y in range(1, 6)]
    assert cvlike.cv_line_pixels(20, 20, (25, 3), (30, 9)) == []                       # outside: clipped away
    assert cvlike.cv_line_pixels(20, 20, (15, 5), (25, 5)) == [(x, 5) for x in range(15, 20)]
    # a triangle: every outline pixel is set, the rows are single runs, the apex row holds the apex only
    tri = [(10, 1), (18, 14), (3, 9)]
    m = cvlike.cv_fill_convex_poly(np.zeros((16, 22)), tri)
    for a, b in zip(tri, tri[1:] + tri[:1]):
        for x, y in cvlike.cv_line_pixels(22, 16, a, b):
            assert m[y, x] == 1
    _spans_of_mask(m)
    assert m[1].sum() == 1 and m[1, 10] == 1 and m[0].sum() == 0 and m[15].sum() == 0
    # a one-row polygon and two points: the outline alone
    assert cvlike.cv_fill_convex_poly(np.zeros((5, 9)), [(1, 2), (7, 2), (4, 2)]).sum() == 7
    assert cvlike.cv_fill_convex_poly(np.zeros((5, 9)), [(1, 1), (3, 3)]).sum() == 3


def test_closed_forms_equal_the_literal_restatement(tmp_path):
    emu = h.hostemu()
    n_checked = 0
    for cam, H, W, seed in ((h.KITTI, 375, 1242, 3000), (h.NUSCENES, 450, 800, 5300), (h.CITYSCAPES, 256, 512, 5100)):
        sc = h.Scene(tmp_path / ('s%d' % H), H, W, 2500, cam=cam, seed0=seed, far_fraction=0.2)
        drops, p64, n64, p32, n32, used, ratio, off = _polygons(sc, 0)
        He, We = sc.He, sc.We
        cvlike.set_fill_rule('cv', int(sc.cam.n_fov))
        try:
            rng = np.random.RandomState(H)
            for k in rng.permutation(len(drops))[:140]:
                n = int(n64[k])
                if n <= 0:
                    continue
                px, py = np.ascontiguousarray(p64[k, 0, :n]), np.ascontiguousarray(p64[k, 1, :n])
                P = np.stack([px, py], 1)
                xl, xr = np.zeros(He, np.int32), np.zeros(He, np.int32)
                applies = emu.emu_rowspans(h._p(px), h._p(py), n, int(sc.cam.n_fov), He, We, 1, h._p(xl), h._p(xr))
                assert bool(applies) == cvlike.fill_rule_cv_applies(P, He, We)
                assert bool(applies) == (n == sc.cam.n_fov) or not applies       # a wrapping polygon never takes OpenCV's rule
                lit = cvlike.fill_fov_mask(np.zeros((He, We)), P)                # literal (cv rule where it applies)
                ml, mr = _spans_of_mask(lit)
                assert np.array_equal(ml, xl) and np.array_equal(mr, xr), k      # C++ closed form (or the span rule) == literal
                y0, pl, pr = cvlike.fov_rowspans(P, He, We)                      # numpy closed form
                assert np.array_equal(pl, xl[y0:y0 + len(pl)]) and np.array_equal(pr, xr[y0:y0 + len(pr)])
                if applies and n_checked % 7 == 0:                               # Clipper's start vertex / orientation cannot matter
                    for r in (1, n // 2):
                        for Q in (np.roll(P, r, 0), np.roll(P, r, 0)[::-1]):
                            assert np.array_equal(cvlike.fill_fov_mask_cv(np.zeros((He, We)), Q), lit)
                n_checked += bool(applies)
        finally:
            cvlike.set_fill_rule()
    assert n_checked > 300


def test_how_far_apart_the_two_rules_are(tmp_path):
    """Per drop: texels and colour constants under both rules (KITTI, 375 x 1909 map); then a window of 250 drops composited by
    the numpy oracle under both: same mask, same statuses, the uint8 image within 1 LSB."""
    emu = h.hostemu()
    sc = h.Scene(tmp_path, 375, 1242, 4096, seed0=3000)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0)
    outs = []
    for rule in (0, 1):
        emu.emu_set_fill_rule(rule)
        try:
            outs.append(h.emu_render(sc, bg, bg, env, drops))
        finally:
            emu.emu_set_fill_rule(1)
    a, b = outs
    assert np.array_equal(a['status'], b['status']) and np.array_equal(a['mask'], b['mask'])
    ok = a['status'] == 0
    rel = np.abs(b['K'][ok] - a['K'][ok]) / np.abs(a['K'][ok])
    assert ok.sum() > 2000 and 1e-5 < rel.max() < 4e-3, rel.max()              # they DO differ: the outline adds ~0.3 % texels
    d = np.abs(a['image_u8'].astype(int) - b['image_u8'].astype(int))
    assert d.max() <= 1 and (d != 0).mean() < 0.05, (d.max(), (d != 0).mean())
    assert np.abs(a['rainy_bg'] - b['rainy_bg']).max() < 0.5 / 255                # less than half an LSB before the quantisation
    # the numpy oracle under both rules on a window
    lo, hi = 1000, 1250
    refs = []
    for rule in ('span', 'cv'):
        cvlike.set_fill_rule(rule, int(sc.cam.n_fov))
        try:
            refs.append(h.oracle_render(sc, 0, bg, bg, env, faithful=True, first_drop=lo, max_drops=hi))
        finally:
            cvlike.set_fill_rule()
    assert np.array_equal(refs[0]['mask'], refs[1]['mask']) and np.array_equal(refs[0]['status'], refs[1]['status'])
    assert np.abs(refs[0]['image_u8'].astype(int) - refs[1]['image_u8'].astype(int)).max() <= 1
    # and the host build of the kernel arithmetic agrees with the oracle under OpenCV's rule as it does under the span rule
    emu.emu_set_fill_rule(1)
    try:
        e = h.emu_render(sc, bg, bg, env, drops[lo:hi])
    finally:
        emu.emu_set_fill_rule(1)
    assert np.array_equal(e['mask'], refs[1]['mask'])
    assert np.abs(e['rainy_bg'] - refs[1]['rainy_bg']).max() < 1e-9
