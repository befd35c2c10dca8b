"""GPU tier: hand-built streaks that force the code paths random scenes rarely reach:
integer-ratio INTER_AREA (ResizeAreaFast), the bilinear fallback of INTER_AREA when the
streak is longer than the rotated texture, frame-border crops on every side, footprints
entirely outside the frame, drops beyond the 10 m sphere, heavy defocus, huge textures that
do not fit the LDS staging, and an empty drop list."""
import numpy as np
import pytest

import helpers as h

pytestmark = pytest.mark.gpu

H, W = 400, 300
FPX = 6e-3 / 4.65e-6


def _drop(pid, x0, y0, x1, y1, iw1, iw2, depth):
    """Streak from image (x0,y0) to (x1,y1) in TOP-LEFT pixel coordinates; the XML carries
    bottom-left y (the loader flips it, bad_weather.py:221-222)."""
    X = (x0 - W / 2) * depth / FPX
    Y = ((H - y0) - H / 2) * depth / FPX
    return dict(pid=pid, wp1=(X, Y, -depth), wp2=(X + 0.001, Y - 0.01, -depth + 0.005), wd1=0.002, wd2=0.002,
                ip1=(x0, H - y0), ip2=(x1, H - y1), iw1=iw1, iw2=iw2)


def _frames():
    d = []
    k = 0
    # exactly vertical -> theta = 0 -> canvas 32x320 / 32x229 ...; tw = max_width+2, th = |dy|
    for (dy, iw) in [(160, 2.5), (80, 2.5), (40, 2.5), (320, 1.5), (64, 3.5), (20, 2.2)]:
        d.append(_drop(k, 40 + 12 * k, 20, 40 + 12 * k, 20 + dy, iw, iw, 5.0)); k += 1
    # longer than the texture: bilinear fallback (dst height > canvas height)
    for (dy, dx, iw) in [(380, 0, 1.2), (350, 3, 2.4), (390, -2, 3.2)]:
        d.append(_drop(k, 150 + 10 * k, 5, 150 + 10 * k + dx, 5 + dy, iw, iw, 4.0)); k += 1
    # oblique small/medium streaks
    for (dx, dy, iw) in [(7, 30, 1.5), (-9, 25, 2.7), (25, 26, 3.9), (-30, 12, 1.1), (3, 3, 1.9)]:
        d.append(_drop(k, 100 + 9 * k, 200, 100 + 9 * k + dx, 200 + dy, iw, iw, 6.0)); k += 1
    # Big drops, some heavily defocused (close), crossing every border
    for (x0, y0, dx, dy, iw1, iw2, z) in [(-6, 50, 8, 60, 7.0, 9.0, 0.5), (290, 100, 14, 70, 6.0, 6.5, 0.8), (120, -20, 5, 50, 5.0, 5.5, 1.5),
                                          (200, 380, -4, 40, 8.0, 8.0, 0.3), (60, 300, 0, 45, 4.0, 12.0, 2.5), (150, 150, 20, 5, 10.0, 4.0, 6.0)]:
        d.append(_drop(k, x0, y0, x0 + dx, y0 + dy, iw1, iw2, z)); k += 1
    # start outside top-left with the end inside; start beyond bottom-right with the end inside
    d.append(_drop(k, -10, -12, 4, 6, 2.5, 2.5, 0.4)); k += 1
    d.append(_drop(k, 310, 405, 295, 390, 1.5, 1.5, 0.6)); k += 1
    # beyond the rendering sphere: skipped; just inside: rendered
    d.append(_drop(k, 80, 120, 82, 140, 2.0, 2.0, 10.3)); k += 1
    d.append(_drop(k, 90, 120, 92, 140, 2.0, 2.0, 9.95)); k += 1
    # very close: large circle of confusion
    d.append(_drop(k, 220, 60, 223, 110, 5.0, 6.0, 0.12)); k += 1
    # closer still: blur radius > 48 -> the two-pass global fallback; and a Medium one through the same path
    d.append(_drop(k, 60, 150, 62, 190, 4.5, 4.5, 0.08)); k += 1
    d.append(_drop(k, 250, 250, 251, 262, 2.5, 2.5, 0.09)); k += 1
    return [dict(id=0, t=2000, d=0, drops=d)]


def _run(tmp_path, built, **kw):
    sc = h.Scene(tmp_path, H, W, 0, frames=_frames(), **kw)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0)
    rh = h.hb.RainHip(0)
    rh.set_streak_db(sc.db.streaks_light)
    rh.set_camera(sc.cam)
    out = rh.render_frames([dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops)])[0]
    rh.close()
    ref = h.oracle_render(sc, 0, bg, bg, env, faithful=False)
    emu = h.emu_render(sc, bg, bg, env, drops)
    return sc, drops, out, ref, emu


def _check(out, ref):
    assert np.array_equal(out['status'], ref['status'])
    assert np.array_equal(out['mask'], ref['mask']), np.abs(out['mask'] - ref['mask']).max()
    assert np.array_equal(out['mask_i32'], ref['mask_i32'])
    assert np.abs(out['image_u8'].astype(int) - ref['image_u8'].astype(int)).max() <= 1     # +-1 LSB


def test_forced_paths_match_oracle(tmp_path, built):
    sc, drops, out, ref, emu = _run(tmp_path, built)
    assert len(drops) >= 26
    assert (ref['status'] == 1).sum() >= 1 and (ref['status'] == 0).sum() >= 20          # > radius skipped, rest rendered
    _check(out, ref)
    _check(out, emu)
    assert out['mask'].max() > 0


def test_textures_too_large_for_lds(tmp_path, built):
    """64-wide, up to 640-tall textures exceed the LDS staging: the global-memory sampler path."""
    sc, drops, out, ref, emu = _run(tmp_path, built, tex_heights=(640, 458, 320, 228, 160), tex_width=64)
    _check(out, ref)


def test_empty_drop_list_is_identity(tmp_path, built):
    sc = h.Scene(tmp_path, 64, 96, 10)
    bg, env = sc.frame_inputs(0)
    rh = h.hb.RainHip(0)
    rh.set_streak_db(sc.db.streaks_light)
    rh.set_camera(sc.cam)
    out = rh.render_frames([dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=np.zeros(0, h.hb.DROP_DTYPE))])[0]
    rh.close()
    assert out['mask'].max() == 0 and np.array_equal(out['rainy_bg'], bg)
    assert np.array_equal(out['image_u8'], (np.clip(bg[..., ::-1], 0, 1) * 255).astype(np.uint8))


def test_bad_arguments_are_errors(built):
    rh = h.hb.RainHip(0)
    with pytest.raises(RuntimeError):           # DB / camera not set
        rh.render_frames([dict(bg=np.zeros((8, 8, 3)), rainy_bg=np.zeros((8, 8, 3)), env_xyY=np.zeros((8, 9, 3)),
                               omega=np.ones((8, 9)), drops=np.zeros(0, h.hb.DROP_DTYPE))])
    rh.close()


@pytest.mark.parametrize("Hh,Ww", [(600, 200), (1100, 96)])
def test_tall_environment_maps(tmp_path, built, Hh, Ww):
    """He = 600 uses the 1024-row LDS span tables of the colour kernel, He = 1100 exceeds them and falls
    back to the per-row edge scan; the kitti-shaped tests only reach the 512-row variant."""
    sc = h.Scene(tmp_path, Hh, Ww, 80, seed0=13)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0)
    rh = h.hb.RainHip(0)
    rh.set_streak_db(sc.db.streaks_light)
    rh.set_camera(sc.cam)
    out = rh.render_frames([dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops)])[0]
    rh.close()
    emu = h.emu_render(sc, bg, bg, env, drops)
    assert np.array_equal(out['status'], emu['status'])
    assert np.array_equal(out['mask'], emu['mask'])
    assert np.abs(out['image_u8'].astype(int) - emu['image_u8'].astype(int)).max() <= 1
    assert np.abs(out['rainy_bg'] - emu['rainy_bg']).max() < 1e-9


def test_nan_pixels_follow_the_reference_blend(tmp_path, built):
    """The compositor's short blend (exact reciprocal division, hardware clamp) is only taken where every factor is tame;
    a NaN in rainy_bg must survive every blend the way np.clip keeps it (bad_weather.py:443-446) -- the literal
    blend_pixel, compared with its host build value for value.  (Values outside [0, 1] are outside the contract of
    rainy_bg: the reference clips them wherever a drop's all-zero pad covers them; the library visits the pads only with
    RR_OPT_WILD_PIXELS -- the next test.)"""
    H2, W2 = 96, 160
    sc = h.Scene(tmp_path, H2, W2, 400, seed0=77)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0)
    wild = bg.copy()
    rng = np.random.RandomState(5)
    ys, xs = rng.randint(0, H2, 400), rng.randint(0, W2, 400)
    vals = np.array([np.nan, np.nan, 1e-300, -0.0, 0.0, 1.0])
    wild[ys, xs, rng.randint(0, 3, 400)] = vals[rng.randint(0, len(vals), 400)]
    rh = h.hb.RainHip(0)
    rh.set_streak_db(sc.db.streaks_light)
    rh.set_camera(sc.cam)
    for rb in (bg, wild):
        out = rh.render_frames([dict(bg=bg, rainy_bg=rb, env_xyY=env, omega=sc.omega, drops=drops)])[0]
        emu = h.emu_render(sc, bg, rb, env, drops)
        assert np.array_equal(out['mask'], emu['mask'])
        a, b = out['rainy_bg'], emu['rainy_bg']
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.isnan(b).sum() == np.isnan(rb).sum()
        fin = ~np.isnan(b)
        # colour constants come from FOV sums added in another order: 1e-9; everything else is the same arithmetic
        assert np.abs(a[fin] - b[fin]).max() < 1e-9
        # the float-colour compositor (no float64 composite requested): a wave that meets such pixels blends them the
        # literal way too; the mask is the same float64 sum
        out32 = rh.render_frames([dict(bg=bg, rainy_bg=rb, env_xyY=env, omega=sc.omega, drops=drops)], want_composite=False)[0]
        assert np.array_equal(out32['mask'], emu['mask']) and np.array_equal(out32['status'], out['status'])
        if rb is bg:
            assert np.abs(out32['image_u8'].astype(int) - emu['image_u8'].astype(int)).max() <= 1
    rh.close()


def test_wild_pixel_values_follow_the_reference_blend(tmp_path, built):
    """rainy_bg from a third party may hold anything.  The reference blends a drop over its whole padded rectangle
    (bad_weather.py:429-446): over the all-zero pad that is np.clip(pixel, 0, 1) -- +-inf, 1e60, -3, 7.5 are clipped wherever
    some drop's pad covers them, before or without any real blend.  With RR_OPT_WILD_PIXELS the library tracks the pads
    (k_pad_visits) and reproduces it: the float64 composite against the host build of the same blend, which walks every
    drop's padded rectangle the way the reference does -- value for value, NaN for NaN."""
    H2, W2 = 96, 160
    sc = h.Scene(tmp_path, H2, W2, 400, seed0=77)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0)
    rng = np.random.RandomState(6)
    ys, xs, cs = rng.randint(0, H2, 1500), rng.randint(0, W2, 1500), rng.randint(0, 3, 1500)
    any_vals = np.array([np.nan, np.inf, -np.inf, 1e60, -1e60, -3.0, 7.5, 1e-300, -0.0, 1.0])
    fin_vals = np.array([-3.0, 7.5, 1.25, -1e-9, 1.0 + 1e-12, 1e-300, -0.0, 1.0])
    wild_any, wild_fin = bg.copy(), bg.copy()
    wild_any[ys, xs, cs] = any_vals[rng.randint(0, len(any_vals), 1500)]
    wild_fin[ys, xs, cs] = fin_vals[rng.randint(0, len(fin_vals), 1500)]
    rh = h.hb.RainHip(0)
    rh.set_streak_db(sc.db.streaks_light)
    rh.set_camera(sc.cam)
    rh.set_option(h.hb.RR_OPT_WILD_PIXELS, 1)
    differs_without = False
    for rb in (bg, wild_fin, wild_any):
        fr = dict(bg=bg, rainy_bg=rb, env_xyY=env, omega=sc.omega, drops=drops)
        out = rh.render_frames([fr])[0]
        emu = h.emu_render(sc, bg, rb, env, drops)
        assert np.array_equal(out['mask'], emu['mask']) and np.array_equal(out['status'], emu['status'][:len(out['status'])])
        a, b = out['rainy_bg'], emu['rainy_bg']
        assert np.array_equal(np.isnan(a), np.isnan(b))
        assert np.array_equal(np.isinf(a), np.isinf(b)) and np.array_equal(a[np.isinf(a)], b[np.isinf(b)])
        fin = np.isfinite(b)
        # colour constants come from FOV sums added in another order: 1e-9 (relative, for the 1e60s no pad reaches)
        assert (np.abs(a[fin] - b[fin]) <= 1e-9 * np.maximum(1.0, np.abs(b[fin]))).all()
        out32 = rh.render_frames([fr], want_composite=False)[0]                  # the float-colour compositor clips first too
        assert np.array_equal(out32['mask'], emu['mask'])
        if rb is not wild_any:                                                    # (an inf or NaN left in the composite poisons the mean shift)
            assert np.abs(out['image_u8'].astype(int) - emu['image_u8'].astype(int)).max() <= 1
            assert np.abs(out32['image_u8'].astype(int) - emu['image_u8'].astype(int)).max() <= 1
        if rb is wild_fin:                                                        # ... and against the numpy oracle itself, which blends every
            ref = h.oracle_render(sc, 0, bg, rb, env, faithful=True)             # drop over its whole padded rectangle like the reference
            assert np.array_equal(out['mask'], ref['mask']) and np.array_equal(out['status'], ref['status'][:len(out['status'])])
            assert (np.abs(a - ref['rainy_bg']) <= 2e-9 * np.maximum(1.0, np.abs(ref['rainy_bg']))).all()
            assert np.abs(out['image_u8'].astype(int) - ref['image_u8'].astype(int)).max() <= 1
            assert np.abs(out32['image_u8'].astype(int) - ref['image_u8'].astype(int)).max() <= 1
        if rb is wild_fin:                                                        # the pads matter: without the option the composite differs
            rh.set_option(h.hb.RR_OPT_WILD_PIXELS, 0)
            plain = rh.render_frames([fr])[0]
            rh.set_option(h.hb.RR_OPT_WILD_PIXELS, 1)
            differs_without = np.abs(plain['rainy_bg'] - b).max() > 0.1
            assert np.array_equal(plain['mask'], emu['mask'])
    assert differs_without
    rh.close()


def test_collinear_polygons_are_skipped(tmp_path, built, monkeypatch):
    """pyclipper's AddPath (bad_weather.py:368) raises for a path without three non-collinear vertices and the reference's
    caller skips the drop.  A cone of a fraction of a degree makes the truncated polygons cover a texel or two: every way
    the library evaluates a polygon -- the thread-per-drop float kernel (slivers go to float64 through the frame's list),
    the edge-parallel kernel with float and with float64 vertices, the general path -- gives the oracle's statuses and mask."""
    import test_hostemu_vs_oracle as tho
    sc = tho.sliver_scene(tmp_path, monkeypatch)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0)
    ref = h.oracle_render(sc, 0, bg, bg, env, faithful=True)
    st = ref['status']
    assert (st == h.orc.ST_FOV_FAIL).sum() >= 10 and (st == 0).sum() >= 10
    fr = dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops)
    for opts, composite in (((), False), ((), True), (((h.hb.RR_OPT_FOV_DDA, 0),), False), (((h.hb.RR_OPT_GENERAL_FOV, 1),), False),
                            (((h.hb.RR_OPT_FOV_F32, 0),), False)):
        rh = h.hb.RainHip(0)
        try:
            rh.set_streak_db(sc.db.streaks_light)
            rh.set_camera(sc.cam)
            for o, v in opts:
                rh.set_option(o, v)
            out = rh.render_frames([fr], want_composite=composite)[0]
        finally:
            rh.close()
        _check(out, ref)
