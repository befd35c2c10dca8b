"""GPU tier: RR_OPT_DEPTH_OCCLUSION -- the depth-buffer occlusion test north_star names (the reference only sketches it:
common/drop_depth_map.py is dead code behind USE_DEPTH_WEIGHTING = 0).  A NEW feature, default off, excluded from the
parity runs; what can be checked is its definition: a drop farther than the scene at a pixel is not composited there."""
import numpy as np
import pytest

import helpers as h

pytestmark = pytest.mark.gpu


def test_depth_occlusion_option(tmp_path, built):
    H, W = 128, 256
    sc = h.Scene(tmp_path, H, W, 400, seed0=21)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0)
    z = np.abs(drops['wps'][:, 2])
    rh = h.hb.RainHip(0)
    try:
        rh.set_streak_db(sc.db.streaks_light)
        rh.set_camera(sc.cam)
        fr = dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops)
        base = rh.render_frames([fr])[0]
        far = np.full((H, W), 1.0e6, np.float32)
        # option off: a depth buffer is ignored
        off = rh.render_frames([dict(fr, depth=np.full((H, W), 0.1, np.float32))])[0]
        assert np.array_equal(off['mask'], base['mask']) and np.array_equal(off['image_u8'], base['image_u8'])
        rh.set_option(h.hb.RR_OPT_DEPTH_OCCLUSION, 1)
        # nothing in front of the drops: unchanged
        same = rh.render_frames([dict(fr, depth=far)])[0]
        assert np.array_equal(same['mask'], base['mask']) and np.array_equal(same['image_u8'], base['image_u8'])
        # a wall at d0 metres hides every drop beyond it == rendering only the drops in front of it
        d0 = float(np.median(z))
        wall = rh.render_frames([dict(fr, depth=np.full((H, W), d0, np.float64))])[0]
        rh.set_option(h.hb.RR_OPT_DEPTH_OCCLUSION, 0)
        near = rh.render_frames([dict(fr, drops=drops[z <= d0])])[0]
        assert 50 < (z <= d0).sum() < len(drops) - 50
        assert np.array_equal(wall['mask'], near['mask']) and np.array_equal(wall['rainy_bg'], near['rainy_bg'])
        assert np.array_equal(wall['image_u8'], near['image_u8'])
        assert wall['mask'].sum() < base['mask'].sum()
        # per pixel: wall only in the left half
        rh.set_option(h.hb.RR_OPT_DEPTH_OCCLUSION, 1)
        half = far.astype(np.float64)                         # (float64: d0 is one of the drops' own distances)
        half[:, :W // 2] = d0
        mixed = rh.render_frames([dict(fr, depth=half)])[0]
        assert np.array_equal(mixed['mask'][:, :W // 2], near['mask'][:, :W // 2])
        assert np.array_equal(mixed['mask'][:, W // 2:], base['mask'][:, W // 2:])
    finally:
        rh.close()
