"""CPU tier: how close a Big drop's warp coordinates sit to a rounding boundary.

cv2.warpPerspective quantises every destination pixel's source coordinate to 1/32 pixel: X = saturate_round(fx * 32 / w)
(generator.py:129-131 -> warpPerspectiveInvoker; rr_device.h warp_big_pixel).  The 3 x 3 matrix comes from
getPerspectiveTransform, which OpenCV 3.2 solves with an SVD where this build (and OpenCV 4) eliminates with partial pivoting
(solve8) -- both backward stable on these well-conditioned 8 x 8 systems, so their matrices differ by a few units in the last
place.  This test perturbs every entry of the inverse matrix the plan holds by +-4 ulp (16 sign patterns + all up / all
down) and counts the pixels of every Big tile whose X or Y changes: the fraction of Big-drop pixels within rounding distance
of the other solver's result.  A changed coordinate moves one bicubic tap table entry (1/32 px): an alpha change of the
order of 1e-3 of the texture's local contrast -- it could move a float64 mask bit, which is why README does not call the Big
branch bit-exact against a real cv2, only against the oracle."""
import ctypes

import numpy as np
import pytest

import helpers as h

PLAN_DTYPE = np.dtype([(k, '<i4') for k in ('status kind tex flip tw th shift pw ph r1 r2 vis_x0 vis_y0 vis_w vis_h crop_x crop_y ew bw0 nW nH '
                                            'rs_mode isx isy eh epitch epad').split()] + [('pad_', '<i4'), ('a0_off', '<i8'), ('a1_off', '<i8'),
                      ('sig1', '<f8'), ('sig2', '<f8'), ('tau_one', '<f8'), ('g', '<f8'), ('mi', '<f8', (9,)), ('ma', '<f8', (6,)),
                      ('scale_x', '<f8'), ('scale_y', '<f8'), ('inv_sx', '<f8'), ('inv_sy', '<f8')])


def _coords(mi, tw, th, bw0):
    """X, Y (the 1/32-pixel integers) of every pixel of a tw x th Big tile: the arithmetic of rr_device.h warp_big_pixel."""
    y, x = np.mgrid[0:th, 0:tw].astype(np.float64)
    bx = np.floor(x / bw0) * bw0
    x1 = x - bx
    X0 = mi[0] * bx + mi[1] * y + mi[2]
    Y0 = mi[3] * bx + mi[4] * y + mi[5]
    W0 = mi[6] * bx + mi[7] * y + mi[8]
    Wd = W0 + mi[6] * x1
    with np.errstate(divide='ignore', invalid='ignore'):
        Wi = np.where(Wd != 0.0, 32.0 / Wd, 0.0)
    fX = np.clip((X0 + mi[0] * x1) * Wi, -2147483648.0, 2147483647.0)
    fY = np.clip((Y0 + mi[3] * x1) * Wi, -2147483648.0, 2147483647.0)
    return np.rint(fX).astype(np.int64), np.rint(fY).astype(np.int64)


def _nudge(m, signs, ulps):
    out = m.copy()
    for _ in range(ulps):
        out = np.nextafter(out, np.where(signs > 0, np.inf, -np.inf))
    return out


def big_drop_rounding_census(sc, frame=0, ulps=4):
    emu = h.hostemu()
    assert emu.emu_sizeof_plan() == PLAN_DTYPE.itemsize
    texels, hs, ws, offs = h.hb.pack_streak_db(sc.db.streaks_light)
    drops = np.ascontiguousarray(sc.product_drops(frame))
    n = len(drops)
    plans = np.zeros(n, PLAN_DTYPE)
    poly, npts, sizes = np.zeros(n * 72, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int64)
    emu.emu_plan(h._p(drops), n, ctypes.byref(sc.cam), sc.H, sc.W, sc.He, sc.We, h._p(hs), h._p(ws), ctypes.c_double(1.0), h._p(plans),
                 h._p(poly), h._p(npts), h._p(sizes))
    big = np.nonzero((plans['kind'] == 0) & (plans['status'] == 0) & (sizes > 0))[0]
    rng = np.random.RandomState(7)
    patterns = [np.ones(9), -np.ones(9)] + [rng.choice([-1.0, 1.0], 9) for _ in range(16)]
    tot = moved = tiles_hit = 0
    for k in big:
        p = plans[k]
        X, Y = _coords(p['mi'], int(p['tw']), int(p['th']), int(p['bw0']))
        hit = np.zeros(X.shape, bool)
        for s in patterns:
            Xp, Yp = _coords(_nudge(p['mi'], s, ulps), int(p['tw']), int(p['th']), int(p['bw0']))
            hit |= (Xp != X) | (Yp != Y)
        tot += hit.size
        moved += int(hit.sum())
        tiles_hit += bool(hit.any())
    return dict(big_drops=len(big), pixels=tot, pixels_at_risk=moved, tiles_with_a_pixel_at_risk=tiles_hit)


@pytest.mark.parametrize("name,cam,H,W,N", [('kitti_100', h.KITTI, 375, 1242, 8192), ('cityscapes_half', h.CITYSCAPES, 512, 1024, 4096),
                                            ('nuscenes_100', h.NUSCENES, 900, 1600, 6000)])
def test_big_drop_pixels_within_rounding_distance(tmp_path, name, cam, H, W, N):
    sc = h.Scene(tmp_path, H, W, N, cam=cam, seed0=4000)
    c4 = big_drop_rounding_census(sc, ulps=4)
    print(name, c4)
    assert c4['big_drops'] > 100 and c4['pixels'] > 20000
    # At +-4 ulp a coordinate flips only where fx * 32 / w lies within ~1e-12 of a half-integer.  That is not a random event: it
    # happens in the few tiles whose quad maps texture rows / columns onto exact half steps of the 1/32 grid (a streak whose end
    # points differ by a divisor of the texture size), and there it happens for whole rows of pixels.  Measured (seed 4000):
    # KITTI 100 mm/hr 4 of 1105 Big tiles / 1382 of 598 892 Big-drop pixels (0.23 %), Cityscapes (render scale 2) 1 pixel of 933 295,
    # nuScenes 100 mm/hr 11 of 918 tiles / 104 of 2 922 822 pixels; the bounds leave room for other seeds.
    assert c4['tiles_with_a_pixel_at_risk'] <= 0.02 * c4['big_drops'], c4
    assert c4['pixels_at_risk'] <= 0.01 * c4['pixels'], c4
