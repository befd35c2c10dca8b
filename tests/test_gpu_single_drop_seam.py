"""GPU tier: the reference's inner seams with their own signatures.  RainRenderer.add_drop_to_image
(bad_weather.py:336-338) takes a caller-made tile, its position and its field-of-view polygon; called drop by drop with the
tiles and polygons the reference's compute_drop would hand over (here: the oracle's restatements of them), it must leave
the very mask and image the batched call produces."""
import copy

import numpy as np
import pytest

import helpers as h
from oracle import render as orc

pytestmark = pytest.mark.gpu


def test_add_drop_to_image_wrapper_equals_batched_call(tmp_path, built):
    H, W, n = 96, 160, 60
    sc = h.Scene(tmp_path, H, W, 150, seed0=10, far_fraction=0.1)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0)[:n]
    rh = h.hb.RainHip(0)
    rh.set_streak_db(sc.db.streaks_light)
    rh.set_camera(sc.cam)
    batched = rh.render_frames([dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops)])[0]
    rh.close()
    assert (batched['status'] != 0).any() and (batched['status'] == 0).sum() > 40

    textures, ratio = sc.oracle_db()
    streaks = sc.oracle_streaks(0)[:n]
    renderer = h.bw.RainRenderer(focal=sc.ocam['focal_m'], f_number=sc.ocam['f_number'], focus_plane=6, radius=10, fov=165)
    fov = h.bw.FovComputation(camera=np.array([0, 0, 0]))
    rainy_bg, rainy_mask, sat = bg.copy(), np.zeros((H, W)), np.zeros((H, W, 3))
    o_bg, o_mask = bg.copy(), np.zeros((H, W))                          # the oracle's own run of the same seam, drop by drop
    fc = orc.FrameConsts(env, sc.omega)
    rain_layer = np.zeros((H, W, 4))
    np.random.seed(0)
    skipped = []
    for i, d in enumerate(streaks):
        tex_idx = orc.take_drop_texture_index(d, ratio)               # the draws of compute_drop (generator.py:119-136)
        if d.drop_type != orc.DropType.Big:
            np.random.normal(0.0, 0.0)
        dd = copy.deepcopy(d)
        tile, minC = orc.make_drop_tile(dd, textures[tex_idx], 0.0, W, H)
        pts, _, _, _ = fov.compute_fov_plane_points(dd, 10, 165, 20, env.shape)
        try:
            # unpacked exactly as the reference's caller does (generator.py:180-183)
            rainy_bg_r, rainy_mask_r, sat_r, drop_vis, blended_drop, minC_out = \
                renderer.add_drop_to_image('kitti', env, sc.omega, pts, minC, bg, rainy_bg, rainy_mask, sat, tile, dd,
                                           'ambient', None, 1.0)
            assert rainy_bg_r is rainy_bg and rainy_mask_r is rainy_mask and sat_r is sat   # in place and returned
            skipped.append(0)
        except Exception:                                                # generator.py:180-189: any exception == skip
            skipped.append(1)
            continue
        # the three tile outputs (bad_weather.py:462) against the oracle's restatement of the same call
        o_vis, o_blend, o_minC = orc.add_drop_to_image(env, sc.omega, fc, pts, minC, bg.shape, o_bg, o_mask, tile, copy.deepcopy(d),
                                                       sc.ocam, 1.0, True, None)
        assert np.array_equal(np.asarray(minC_out), np.asarray(o_minC))
        assert drop_vis.shape == o_vis.shape and blended_drop.shape == o_blend.shape == drop_vis.shape[:2] + (3,)
        assert np.array_equal(drop_vis[..., 3], o_vis[..., 3])          # alpha: bit-exact
        assert np.abs(drop_vis[..., :3] - o_vis[..., :3]).max() < 1e-12  # colour: K x blurred alpha vs blurred (K x alpha)
        assert np.abs(blended_drop - o_blend).max() < 1e-12
        # and what the reference's caller does with them next (generator.py:437-438) runs
        rain_layer = renderer.make_rain_layer(drop_vis, blended_drop, rain_layer, rainy_mask, minC_out)
    assert rain_layer[..., 3].max() == 255 and np.array_equal(rain_layer[..., 3] > 0, rainy_mask > 0)
    assert np.array_equal(np.array(skipped), (batched['status'] != 0).astype(int))
    assert np.array_equal(rainy_mask, batched['mask'])                  # bit-exact
    assert np.abs(rainy_bg - batched['rainy_bg']).max() < 1e-12
    assert sat.max() == 0                                               # dead output: passed through

    # 'white' strategy through the same seam: no polygon, no defocus
    rh = h.hb.RainHip(0)
    rh.set_streak_db(sc.db.streaks_light)
    rh.set_camera(sc.cam)
    ref_w = rh.render_frames([dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops, strategy=1)])[0]
    rh.close()
    rb, rm = bg.copy(), np.zeros((H, W))
    np.random.seed(0)
    for d in streaks:
        tex_idx = orc.take_drop_texture_index(d, ratio)
        if d.drop_type != orc.DropType.Big:
            np.random.normal(0.0, 0.0)
        dd = copy.deepcopy(d)
        tile, minC = orc.make_drop_tile(dd, textures[tex_idx], 0.0, W, H)
        _, _, _, vis_w, blend_w, minC_w = renderer.add_drop_to_image('kitti', env, sc.omega, np.array([]), minC, bg, rb, rm, sat, tile, dd,
                                                                     'ambient', 'white', 1.0)
        assert np.array_equal(np.asarray(minC_w), np.asarray(minC)) and vis_w.shape[:2] == blend_w.shape[:2]
        assert np.array_equal(vis_w, tile[:vis_w.shape[0], :vis_w.shape[1]])
    assert np.array_equal(rm, ref_w['mask']) and np.array_equal(rb, ref_w['rainy_bg'])
