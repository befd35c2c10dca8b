"""CPU tier: the float32 field-of-view polygon of the colour branch (rr_device.h fov_polygon_auto, what k_fov_spans evaluates
by default) against the float64 one, on the host build (glibc's atan2f / sqrtf in place of the device's).  The polygon only
scales rainy_image (+-1 LSB); what it must never change is a drop's STATUS or the polygon's SHAPE: every predicate that
decides those carries an error bound, and a drop that comes within it is evaluated in float64."""
import ctypes

import numpy as np
import pytest

import helpers as h


def _polygons(sc, i):
    emu = h.hostemu()
    texels, hs, ws, offs = h.hb.pack_streak_db(sc.db.streaks_light)
    drops = np.ascontiguousarray(sc.product_drops(i))
    n = len(drops)
    plans = np.zeros(n * emu.emu_sizeof_plan(), np.uint8)
    p64, n64, sizes = np.zeros(n * 72, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int64)
    emu.emu_plan(h._p(drops), n, ctypes.byref(sc.cam), sc.H, sc.W, sc.He, sc.We, h._p(hs), h._p(ws), ctypes.c_double(1.0), h._p(plans),
                 h._p(p64), h._p(n64), h._p(sizes))
    p32, n32, used = np.zeros(n * 72, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    emu.emu_fov_auto(h._p(drops), n, ctypes.byref(sc.cam), sc.He, sc.We, h._p(p32), h._p(n32), h._p(used))
    ratio, offs_px = np.zeros(n), np.zeros(n)
    emu.emu_fov_error_ratio(h._p(drops), n, ctypes.byref(sc.cam), sc.He, sc.We, h._p(ratio), h._p(offs_px))
    return drops, p64.reshape(n, 2, 36), n64, p32.reshape(n, 2, 36), n32, used, ratio, offs_px


@pytest.mark.parametrize("cam,H,W,N", [(h.KITTI, 375, 1242, 4096), (h.NUSCENES, 450, 800, 2048), (h.CITYSCAPES, 256, 512, 2048)])
def test_float_polygon_keeps_status_and_shape(tmp_path, cam, H, W, N):
    sc = h.Scene(tmp_path, H, W, N, n_frames=2, cam=cam, seed0=5100)
    tot = fall = 0
    for i in range(2):
        drops, p64, n64, p32, n32, used, ratio, off = _polygons(sc, i)
        assert np.array_equal(n32, n64), "vertex count (0 = no polygon: a status) differs"
        keep = (used == 1) & (n32 > 0)
        for k in np.nonzero(keep)[0]:
            assert np.abs(p32[k, :, :n32[k]] - p64[k, :, :n64[k]]).max() <= 1         # a truncated vertex moves by at most one texel
        for k in np.nonzero(used <= 0)[0]:                                              # the float64 polygon, bit for bit
            assert np.array_equal(p32[k, :, :max(n32[k], 0)], p64[k, :, :max(n64[k], 0)])
        ok = ratio >= 0
        assert ratio[ok].max() < 0.5, "azimuth error beyond half of its bound (the margins are 4 bounds wide)"
        assert off[ok].max() < 0.05                                                    # texels
        tot += len(drops)
        fall += int((used <= 0).sum())
    assert len(drops) > 100 and fall < 0.03 * tot, "float64 fall-backs: %d of %d" % (fall, tot)


def test_float_polygon_unsure_cases_go_to_float64(tmp_path):
    """Hand-made drops on the thresholds: beyond the sphere (certain failure, decided in float), on the sphere (float64
    decides), straight above the camera (a vertex circle around the pole), mid-point without depth (b == 0 branch)."""
    emu = h.hostemu()
    sc = h.Scene(tmp_path, 96, 160, 8)
    d = np.zeros(5, h.hb.DROP_DTYPE)
    for k, (p, q) in enumerate([((0.5, 0.3, 30.0), (0.5, 0.2, 30.0)),          # 30 m away: outside the 10 m sphere
                                ((0.0, 0.0, 10.0), (0.0, 0.0, 10.0)),          # on the sphere
                                ((0.001, 5.0, 0.002), (0.001, 5.0, 0.002)),    # straight up
                                ((2.0, 0.5, 0.0), (2.0, 0.5, 0.0)),            # no depth
                                ((0.4, 0.2, 3.0), (0.4, 0.1, 3.0))]):          # an ordinary drop
        d['wps'][k], d['wpe'][k] = p, q
    n = len(d)
    p64, n64 = np.zeros(n * 72, np.int32), np.zeros(n, np.int32)
    plans, sizes = np.zeros(n * emu.emu_sizeof_plan(), np.uint8), np.zeros(n, np.int64)
    texels, hs, ws, offs = h.hb.pack_streak_db(sc.db.streaks_light)
    emu.emu_plan(h._p(d), n, ctypes.byref(sc.cam), sc.H, sc.W, sc.He, sc.We, h._p(hs), h._p(ws), ctypes.c_double(1.0), h._p(plans), h._p(p64),
                 h._p(n64), h._p(sizes))
    p32, n32, used = np.zeros(n * 72, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    emu.emu_fov_auto(h._p(d), n, ctypes.byref(sc.cam), sc.He, sc.We, h._p(p32), h._p(n32), h._p(used))
    assert np.array_equal(n32, n64)
    assert n64[0] == 0 and used[0] == 1                     # certain failure needs no float64
    assert used[1] <= 0 and used[3] <= 0                    # thresholds: float64
    assert used[4] == 1 and n32[4] in (20, 24)
    for k in range(n):
        if used[k] <= 0:
            assert np.array_equal(p32.reshape(n, 2, 36)[k], p64.reshape(n, 2, 36)[k])


def _dda_bad(px, py, He, We):
    """rows on which the cursors of k_fov_dda (DdaCursors) differ from the rule they walk -- the span rule (fov_rowspan) and,
    where it applies (every vertex on the map), OpenCV's (fov_rowspan_cv, the default since round 6); -1: not monotone"""
    emu = h.hostemu()
    px, py = np.ascontiguousarray(px, np.int32), np.ascontiguousarray(py, np.int32)
    a = emu.emu_dda_check(h._p(px), h._p(py), len(px), He, We, 0)
    b = emu.emu_dda_check(h._p(px), h._p(py), len(px), He, We, 1)
    return a if a < 0 else a + max(b, 0)


def test_thread_per_drop_spans_equal_the_rule(tmp_path):
    """k_fov_dda's two cursors (rr_device.h DdaCursors) against the rules they implement (fov_rowspan: min / max over every
    edge that touches the row; fov_rowspan_cv: OpenCV's outline + edge walk): the polygons of a KITTI and a wide-angle scene, and hand-made ones with flat tops and
    bottoms, repeated vertices, a single row, vertices on the map's border rows."""
    He, We = 375, 1909
    n_checked = n_general = 0
    for cam, H, W in ((h.KITTI, 375, 1242), (h.NUSCENES, 450, 800)):
        sc = h.Scene(tmp_path / ('s%d' % H), H, W, 3000, cam=cam, seed0=5300, far_fraction=0.2)
        drops, p64, n64, p32, n32, used, ratio, off = _polygons(sc, 0)
        for P, Nn in ((p64, n64), (p32, n32)):
            for k in range(len(drops)):
                if Nn[k] > 0:
                    bad = _dda_bad(P[k, 0, :Nn[k]], P[k, 1, :Nn[k]], sc.He, sc.We)
                    if bad < 0:
                        n_general += 1                           # not monotone: the kernel's list (wrapping polygons)
                    else:
                        assert bad == 0, (k, P[k, :, :Nn[k]])
                        n_checked += 1
                    if Nn[k] == 20:
                        assert bad == 0, "a 20-gon that is not monotone: %r" % (P[k, :, :20],)
    assert n_checked > 4000 and n_general < 0.05 * n_checked
    rng = np.random.RandomState(5)
    for trial in range(400):                                     # random monotone polygons, many ties
        n_l, n_r = rng.randint(1, 9), rng.randint(1, 9)
        y_top, y_bot = sorted(rng.randint(0, He + 1, 2))
        ys_l = np.sort(rng.randint(y_top, y_bot + 1, n_l))
        ys_r = np.sort(rng.randint(y_top, y_bot + 1, n_r))[::-1]
        xs_l, xs_r = rng.randint(0, We // 2, n_l), rng.randint(We // 2, We + 1, n_r)
        px = np.concatenate([[rng.randint(0, We)], xs_r[::-1] if False else xs_l, [rng.randint(0, We)], xs_r])
        py = np.concatenate([[y_top], ys_l, [y_bot], ys_r])
        roll = rng.randint(0, len(px))                           # the top vertex anywhere in the loop
        bad = _dda_bad(np.roll(px, roll), np.roll(py, roll), He, We)
        assert bad == 0, (trial, px, py)
    assert _dda_bad([5, 90, 40], [7, 7, 7], He, We) == 0        # all on one row
    assert _dda_bad([5, 9, 9, 5], [0, 0, He, He], He, We) == 0   # border rows (row He lies outside the map)
    assert _dda_bad([3, 3, 8, 8, 8], [2, 2, 2, 9, 9], He, We) == 0   # repeated vertices
    assert _dda_bad([0, 50, 20, 70, 10], [0, 40, 10, 40, 80], He, We) == -1   # up and down twice: not for the cursors
