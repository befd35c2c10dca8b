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
            out = renderer.add_drop_to_image('kitti', env, sc.omega, pts, minC, bg, rainy_bg, rainy_mask, sat, tile, dd,
                                             'ambient', None, 1.0)
            assert out[0] is rainy_bg and out[1] is rainy_mask and out[3] is None      # in place and returned
            skipped.append(0)
        except Exception:                                                # generator.py:180-189: any exception == skip
            skipped.append(1)
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
        renderer.add_drop_to_image('kitti', env, sc.omega, np.array([]), minC, bg, rb, rm, sat, tile, dd, 'ambient', 'white', 1.0)
    assert np.array_equal(rm, ref_w['mask']) and np.array_equal(rb, ref_w['rainy_bg'])
