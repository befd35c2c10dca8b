"""GPU tier: RR_OPT_DEPTH_OCCLUSION -- the depth-buffer occlusion test north_star names (the reference only sketches it:
common/drop_depth_map.py is dead code behind USE_DEPTH_WEIGHTING = 0).  A NEW feature, default off, excluded from the
parity runs.  Its definition lives OUTSIDE the library: oracle/render.py (_visible: a drop farther from the camera than
the scene at a pixel is neither blended nor added to the mask there) and, independently coded, tests/hostemu; the HIP
option is compared against both on a depth buffer that cuts through the drops' own distance range."""
import numpy as np
import pytest

import helpers as h

pytestmark = pytest.mark.gpu


def _scene_depth(H, W, z, dtype):
    """A ground-plane-like ramp through the drops' distance range plus two walls: most drops are hidden in a part of
    their footprint only."""
    lo, hi = float(np.percentile(z, 10)), float(np.percentile(z, 90))
    depth = np.linspace(hi, lo, H)[:, None] * np.ones((1, W))
    depth[:, W // 3:W // 3 + 20] = lo * 0.5                   # a near wall
    depth[H // 4:H // 4 + 10, :] = 1.0e6                       # a gap to infinity
    depth[0, 0] = np.nan                                       # NaN hides nothing
    return depth.astype(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_depth_occlusion_equals_its_definition(tmp_path, built, dtype):
    H, W = 128, 256
    sc = h.Scene(tmp_path, H, W, 400, seed0=21)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0)
    z = np.abs(drops['wps'][:, 2])
    depth = _scene_depth(H, W, z, dtype)
    rh = h.hb.RainHip(0)
    try:
        rh.set_streak_db(sc.db.streaks_light)
        rh.set_camera(sc.cam)
        fr = dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops)
        base = rh.render_frames([fr])[0]
        # option off: a depth buffer is ignored
        off = rh.render_frames([dict(fr, depth=depth)])[0]
        assert np.array_equal(off['mask'], base['mask']) and np.array_equal(off['image_u8'], base['image_u8'])
        rh.set_option(h.hb.RR_OPT_DEPTH_OCCLUSION, 1)
        got = rh.render_frames([dict(fr, depth=depth)])[0]
        # nothing in front of the drops: unchanged
        same = rh.render_frames([dict(fr, depth=np.full((H, W), 1.0e6, dtype))])[0]
        assert np.array_equal(same['mask'], base['mask']) and np.array_equal(same['image_u8'], base['image_u8'])
    finally:
        rh.close()
    assert 0.2 * base['mask'].sum() < got['mask'].sum() < 0.9 * base['mask'].sum()      # the buffer really hides things
    # 1. the numpy statement of the rule (op-for-op oracle of the reference path + _visible)
    ref = h.oracle_render(sc, 0, bg, bg, env, faithful=True, scene_depth=depth)
    assert np.array_equal(got['status'], ref['status'])
    assert np.array_equal(got['mask'], ref['mask']) and np.array_equal(got['mask_i32'], ref['mask_i32'])      # bit-exact
    assert np.abs(got['image_u8'].astype(int) - ref['image_u8'].astype(int)).max() <= 1                      # +-1 LSB
    assert np.abs(got['rainy_bg'] - ref['rainy_bg']).max() < 1e-12
    # 2. the independent C++ statement (tests/hostemu: the kernels' per-pixel arithmetic under plain loops)
    emu = h.emu_render(sc, bg, bg, env, drops, depth=depth)
    assert np.array_equal(got['mask'], emu['mask']) and np.array_equal(got['image_u8'], emu['image_u8'])


def test_depth_occlusion_wall_equals_dropping_the_far_streaks(tmp_path, built):
    """A property of the rule that needs no second implementation: a wall at d0 metres over the whole frame == rendering only
    the drops in front of it."""
    H, W = 96, 160
    sc = h.Scene(tmp_path, H, W, 300, seed0=22)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0)
    z = np.abs(drops['wps'][:, 2])
    d0 = float(np.median(z))
    rh = h.hb.RainHip(0)
    try:
        rh.set_streak_db(sc.db.streaks_light)
        rh.set_camera(sc.cam)
        fr = dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops)
        near = rh.render_frames([dict(fr, drops=drops[z <= d0])])[0]
        rh.set_option(h.hb.RR_OPT_DEPTH_OCCLUSION, 1)
        wall = rh.render_frames([dict(fr, depth=np.full((H, W), d0, np.float64))])[0]
    finally:
        rh.close()
    assert 30 < (z <= d0).sum() < len(drops) - 30
    assert np.array_equal(wall['mask'], near['mask']) and np.array_equal(wall['rainy_bg'], near['rainy_bg'])
    assert np.array_equal(wall['image_u8'], near['image_u8'])
