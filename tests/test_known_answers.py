"""CPU tier: KNOWN-ANSWER tests of the OpenCV / imutils restatements (oracle/cvlike.py) and of the kernel
arithmetic that mirrors them (rr_device.h through tests/hostemu).

The reference's warp / rotate / resize calls (generator.py:126-171) cannot be pinned by import (no cv2 here),
so a slip in cvlike.py would be mirrored by the kernels and stay green in the parity tests.  These cases have
answers that follow from the DEFINITION of the operation, independent of either implementation:

  * identity / integer-translation homography reproduces (shifts) the texture exactly: the cubic weights at
    fraction 0 are exactly (0, 1, 0, 0);
  * INTER_AREA by an exact factor 2 / 4 / 8 is the box mean (exactly, for dyadic inputs);
  * computeResizeAreaTab weights of every destination cell sum to 1;
  * rotate_bound by 0 deg is the identity, by 90 / 180 deg a permutation of the pixels -- with OpenCV's
    well-known one-pixel offset, because imutils rotates about (w/2, h/2), not ((w-1)/2, (h-1)/2).

imutils centre convention: the released imutils (>= 0.4) computes `(cX, cY) = (w / 2, h / 2)` --
true division under Python 3 -- and this is what cvlike.rotate_bound_geometry and plan_drop use; the `//`
form of very old copies differs by half a pixel for odd sizes and is NOT implemented.
"""
import ctypes

import numpy as np

import helpers as h
from oracle import cvlike

RNG = np.random.RandomState(123)


def _dyadic(shape):
    return RNG.randint(0, 256, shape).astype(np.float64) / 256.0      # sums of <= 2^40 of them are exact


# ---------------------------------------------------------------------------------------------------------
# cvlike: warpPerspective / getPerspectiveTransform
# ---------------------------------------------------------------------------------------------------------
def test_identity_homography_reproduces_the_texture():
    tex = RNG.rand(40, 24)
    out = cvlike.warp_perspective_cubic(tex, np.eye(3), 24, 40)
    assert np.array_equal(out, tex)


def test_integer_translation_homography_shifts_the_texture():
    tex = RNG.rand(30, 20)
    tx, ty = 3, 5
    M = np.array([[1.0, 0, tx], [0, 1.0, ty], [0, 0, 1.0]])
    out = cvlike.warp_perspective_cubic(tex, M, 20 + tx + 2, 30 + ty + 2)
    exp = np.zeros((30 + ty + 2, 20 + tx + 2))
    exp[ty:ty + 30, tx:tx + 20] = tex
    assert np.array_equal(out, exp)                                   # BORDER_CONSTANT 0 outside


def test_perspective_transform_of_known_quads():
    src = np.array([[0, 0], [32, 0], [32, 80], [0, 80]], np.float32)
    assert np.abs(cvlike.get_perspective_transform(src, src) - np.eye(3)).max() < 1e-12
    M = cvlike.get_perspective_transform(src, src + np.float32([7, -3]))
    assert np.abs(M - np.array([[1, 0, 7], [0, 1, -3], [0, 0, 1.0]])).max() < 1e-12
    M = cvlike.get_perspective_transform(src, src * np.float32([0.5, 2.0]))
    assert np.abs(M - np.diag([0.5, 2.0, 1.0])).max() < 1e-12
    # a genuine projective map: the four corners land where they were asked to
    dst = np.array([[1, 2], [30, 0], [33, 79], [-2, 85]], np.float32)
    M = cvlike.get_perspective_transform(src, dst)
    p = np.c_[src.astype(float), np.ones(4)] @ M.T
    assert np.abs(p[:, :2] / p[:, 2:] - dst).max() < 1e-9


# ---------------------------------------------------------------------------------------------------------
# cvlike: INTER_AREA
# ---------------------------------------------------------------------------------------------------------
def test_area_resize_by_exact_factors_is_the_box_mean():
    src = _dyadic((64, 48))
    for fy, fx in [(2, 2), (4, 4), (8, 2), (2, 4), (1, 2), (4, 1)]:
        out = cvlike.resize_area(src, 48 // fx, 64 // fy)
        box = src.reshape(64 // fy, fy, 48 // fx, fx).sum(axis=(1, 3)) / (fx * fy)
        if (fx * fy) & (fx * fy - 1) == 0:
            assert np.array_equal(out, box), (fy, fx)                 # dyadic inputs, power-of-two area: exact
        assert np.abs(out - box).max() < 1e-15


def test_area_resize_fractional_factor_preserves_the_mean_and_weights_sum_to_one():
    for ssize, dsize in [(37, 5), (320, 47), (32, 3), (229, 100), (114, 113)]:
        scale = ssize / dsize
        tab = cvlike._area_tab(ssize, dsize, scale)
        w = np.zeros(dsize)
        cover = np.zeros(ssize)
        for si, di, a in tab:
            w[di] += float(a)
            cover[si] += float(a) * min(scale, ssize - di * scale)     # back to source-pixel units
        assert np.abs(w - 1).max() < 1e-6                               # float32 table entries
        assert np.abs(cover - 1).max() < 1e-5                           # every source pixel is used exactly once
    # constant image -> constant image; mean preserved for a random one
    assert np.abs(cvlike.resize_area(np.full((229, 32), 0.625), 5, 47) - 0.625).max() < 1e-6
    src = RNG.rand(114, 32)
    assert abs(cvlike.resize_area(src, 7, 19).mean() - src.mean()) < 1e-6


def test_area_resize_upsampling_falls_back_to_bilinear():
    """dst larger than src: cv::resize uses INTER_LINEAR with 'area' source coordinates; a 2x up-sampling of a
    horizontal ramp stays inside [min, max], is monotone, and keeps the corner pixels."""
    src = np.tile(np.arange(8.0)[None, :], (4, 1))
    out = cvlike.resize_area(src, 16, 8)
    assert out.shape == (8, 16) and out.min() >= 0 and out.max() <= 7
    assert np.all(np.diff(out, axis=1) >= 0) and np.all(out == out[0])
    assert out[0, 0] == 0 and out[0, -1] == 7


# ---------------------------------------------------------------------------------------------------------
# cvlike: rotate_bound
# ---------------------------------------------------------------------------------------------------------
def test_rotate_bound_by_right_angles_is_a_permutation():
    src = RNG.rand(11, 6)
    h_, w_ = src.shape
    assert np.array_equal(cvlike.rotate_bound(src, 1.0, 0.0), src)                          # 0 deg
    # 180 deg about (w/2, h/2): x' = w - x, y' = h - y  -> pixel (0, 0) has no source (one-pixel offset)
    out = cvlike.rotate_bound(src, -1.0, 0.0)
    exp = np.zeros_like(src)
    exp[1:, 1:] = src[::-1, ::-1][:-1, :-1]
    assert out.shape == src.shape and np.array_equal(out, exp)
    # 90 deg: canvas h x w -> w x h; alpha = 0, beta = -1: x' = h - y, y' = x
    out = cvlike.rotate_bound(src, 0.0, -1.0)
    assert out.shape == (w_, h_)
    exp = np.zeros((w_, h_))
    for yp in range(w_):
        for xp in range(1, h_):
            exp[yp, xp] = src[h_ - xp, yp]
    assert np.array_equal(out, exp)
    # -90 deg: alpha = 0, beta = +1: x' = y, y' = w - x
    out = cvlike.rotate_bound(src, 0.0, 1.0)
    exp = np.zeros((w_, h_))
    for yp in range(1, w_):
        for xp in range(h_):
            exp[yp, xp] = src[xp, w_ - yp]
    assert np.array_equal(out, exp)
    # odd sizes: the centre is (w/2, h/2) = x.5 -- the `w // 2` convention would shift this by half a pixel
    M, nW, nH = cvlike.rotate_bound_geometry(229, 33, 1.0, 0.0)
    assert (nW, nH) == (33, 229) and np.array_equal(M, np.array([[1.0, 0, 0], [0, 1.0, 0]]))


# ---------------------------------------------------------------------------------------------------------
# the kernel arithmetic (rr_device.h, host build) on the same known answers
# ---------------------------------------------------------------------------------------------------------
H, W = 420, 300


def _tile(sc, drop):
    """Raw (un-blurred, in-focus) alpha tile of one hand-made rr_drop through plan_drop + raw_tile_pixel."""
    lib = h.hostemu()
    drops = np.zeros(1, h.hb.DROP_DTYPE)
    for k, v in drop.items():
        drops[k][0] = v
    psz = lib.emu_sizeof_plan()
    plans = np.zeros(psz, np.uint8)
    poly = np.zeros(72, np.int32)
    npts = np.zeros(1, np.int32)
    sizes = np.zeros(1, np.int64)
    texels, hs, ws, offs = h.hb.pack_streak_db(sc.db.streaks_light)
    lib.emu_plan(h._p(drops), 1, ctypes.byref(sc.cam), H, W, sc.He, sc.We, h._p(hs), h._p(ws), ctypes.c_double(1.0),
                 h._p(plans), h._p(poly), h._p(npts), h._p(sizes))
    ints = plans[:24 * 4].view(np.int32)
    tw, th, shift, pw, ph = (int(ints[k]) for k in (4, 5, 6, 7, 8))
    assert shift == 0 and (pw, ph) == (tw, th), "known-answer drops sit on the focus plane (no defocus pad)"
    out = np.zeros((ph, pw))
    lib.emu_tile(h._p(plans), h._p(texels), h._p(hs), h._p(ws), h._p(offs), h._p(out))
    return out


def _base(**kw):
    d = dict(x0=100, y0=20, x1=100, y1=20, max_width=2, length=10, type=1, tex_index=0, iw1=2.5, iw2=2.5,
             wps=(0.0, 0.0, 6.0), wpe=(0.0, -0.01, 6.0), rot_cos=1.0, rot_sin=0.0)       # z = 6 m: circle of confusion 0
    d.update(kw)
    return d


def test_kernel_big_drop_with_identity_quad_reproduces_the_texture(tmp_path):
    """Big branch: d0 = d1 = texture width, vertical extent = texture height -> the destination quad is the
    source quad (up to the reference's own +1e-3 px skew, far below the 1/32 px coordinate grid)."""
    sc = h.Scene(tmp_path, H, W, 4)
    for ti in (0, 25, 49):
        tex = sc.db.streaks_light[ti][..., 0] if sc.db.streaks_light[ti].ndim == 3 else sc.db.streaks_light[ti]
        sh, sw = tex.shape
        out = _tile(sc, _base(type=0, tex_index=ti, x0=50, x1=50, y0=10, y1=10 + sh, iw1=sw + 0.3, iw2=sw + 0.7,
                              max_width=sw, length=sh))
        assert out.shape == (sh, sw)
        assert np.array_equal(out, tex.astype(np.float64) / 255.0)


def test_kernel_area_resize_by_integer_factors_is_the_box_mean(tmp_path):
    """Medium branch with the identity rotation: tile = INTER_AREA(texture) by exactly (isx, isy); flipped
    vertically when the streak ends in the right half of the frame (generator.py:165)."""
    sc = h.Scene(tmp_path, H, W, 4)
    for ti, fx, fy, x_end in [(0, 8, 8, 60), (0, 2, 2, 60), (20, 4, 2, 60), (0, 8, 4, 200), (40, 2, 4, 250)]:
        tex = sc.db.streaks_light[ti][..., 0] if sc.db.streaks_light[ti].ndim == 3 else sc.db.streaks_light[ti]
        sh, sw = tex.shape
        tw, th = sw // fx, sh // fy
        out = _tile(sc, _base(tex_index=ti, x0=x_end - tw, x1=x_end, y0=10, y1=10 + th, max_width=2))
        t = tex.astype(np.float64) / 255.0
        if x_end > W // 2:
            t = t[::-1]
        box = t.reshape(th, fy, tw, fx).mean(axis=(1, 3))
        assert out.shape == (th, tw)
        assert np.abs(out - box).max() < 2e-15, (ti, fx, fy)          # same cells; only the summation order differs
        assert np.array_equal(out, np.clip(cvlike.resize_area(t, tw, th), 0, 1))       # and bit-exact vs cvlike


def test_kernel_rotation_by_right_angles_is_a_permutation(tmp_path):
    """Medium branch, rotate_bound by 180 / 90 degrees followed by a 1:1 'resize': a pure pixel permutation
    with OpenCV's one-pixel offset (see test_rotate_bound_by_right_angles_is_a_permutation)."""
    sc = h.Scene(tmp_path, H, W, 4)
    ti = 30                                                         # 32 x 114 texture
    tex = sc.db.streaks_light[ti][..., 0] if sc.db.streaks_light[ti].ndim == 3 else sc.db.streaks_light[ti]
    t = tex.astype(np.float64) / 255.0
    sh, sw = t.shape
    out = _tile(sc, _base(tex_index=ti, x0=10, x1=10 + sw, y0=10, y1=10 + sh, rot_cos=-1.0, rot_sin=0.0))     # 180 deg
    exp = np.zeros_like(t)
    exp[1:, 1:] = t[::-1, ::-1][:-1, :-1]
    assert np.array_equal(out, exp)
    out = _tile(sc, _base(tex_index=ti, x0=10, x1=10 + sh, y0=10, y1=10 + sw, rot_cos=0.0, rot_sin=-1.0))     # 90 deg
    exp = np.zeros((sw, sh))
    for yp in range(sw):
        exp[yp, 1:] = t[sh - 1:0:-1, yp]
    assert np.array_equal(out, exp)
    assert np.array_equal(out, cvlike.resize_area(cvlike.rotate_bound(t, 0.0, -1.0), sh, sw))


def test_division_by_the_exposure_through_its_reciprocal_is_exact():
    """k_composite's short blend computes (A * tau) / exposure as q0 = a * y, fma(fma(-q0, d, a), y, q0) with y = 1 / d.
    With y correctly rounded that is the correctly rounded quotient (Markstein) unless d's significand is all ones (the
    kernel then divides); checked here with the host's fma on millions of numerators for the cameras' exposures and a
    spread of other divisors."""
    import ctypes
    import os
    import __graft_entry__ as ge
    ge.build()
    emu = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'hostemu', 'libhostemu.so'))
    emu.emu_reciprocal_division_mismatches.restype = ctypes.c_int64
    emu.emu_reciprocal_division_mismatches.argtypes = [ctypes.c_double, ctypes.c_int64, ctypes.c_uint64]
    divisors = [2e-3, 5e-3, 1e-3, 1.0 / 60, 1.0 / 30, 0.0166, 2.5e-3, 0.04, 3.0, 0.75, 1e-6, 123.456, 1.0 / 3, 0.1]
    for k, d in enumerate(divisors):
        assert emu.emu_reciprocal_division_mismatches(d, 3_000_000, k + 1) == 0, d
