"""GPU tier: every BASELINE.json configuration at its own frame size, camera and drop count.

  config 2  KITTI 1242x375, 25 mm/hr                        (config/kitti.py)
  config 3  KITTI 1242x375, 100 mm/hr                       -> also tests/test_gpu_properties.py
  config 4  Cityscapes 2048x1024, 50 mm/hr, 5 ms exposure   (config/cityscapes.py:27-42) and the reference's
            default half-resolution rendering: render_scale = 2 -> 1024x512 (image coordinates / 2 in the loader)
  config 5  nuScenes 1600x900, f = 5.5 mm, f/1.8, 5 ms      (config/nuscenes.py:66-72), {5, 100, 200} mm/hr,
            particles from the generator (no XML file for the product path)

Each frame is rendered through the C ABI and compared
  * at full size with the g++ build of the kernel arithmetic (tests/hostemu): mask bit-exact, image <= 1 LSB;
  * on windows of the drop list with the numpy oracle in its op-for-op ("faithful": per-drop masked reduction
    over the whole environment map) mode: mask bit-exact (f64 and int32), image <= 1 LSB, same skip status.
Tolerances are BASELINE.json's: rainy_mask bit-exact, rainy_image +-1 LSB per channel.
"""
import numpy as np
import pytest

import helpers as h

pytestmark = pytest.mark.gpu

# name -> (H, W, simulated drops, camera, render_scale, oracle windows [(first, last)])
CONFIGS = {
    'kitti_25':        (375, 1242, 2048, h.KITTI, 1, [(0, 500)]),
    'kitti_100':       (375, 1242, 8192, h.KITTI, 1, [(0, 350), (2500, 2850), (6800, 7150)]),
    'cityscapes_half': (512, 1024, 4096, h.CITYSCAPES, 2, [(0, 500)]),
    'cityscapes_full': (1024, 2048, 4096, h.CITYSCAPES, 1, [(0, 300), (1500, 1800), (3000, 3400)]),
    'nuscenes_5':      (900, 1600, 512, h.NUSCENES, 1, [(0, 446)]),
    'nuscenes_100':    (900, 1600, 8192, h.NUSCENES, 1, [(0, 250), (5000, 5250)]),
    'nuscenes_200':    (900, 1600, 16384, h.NUSCENES, 1, [(2000, 2500), (9000, 9500)]),
}


def _assert_parity(out, ref, tag):
    assert np.array_equal(out['status'], ref['status']), tag + ': drop status'
    assert np.array_equal(out['mask'], ref['mask']), tag + ': mask f64, max |d| = %g' % np.abs(out['mask'] - ref['mask']).max()
    assert np.array_equal(out['mask_i32'], ref['mask_i32']), tag + ': mask int32'
    d = np.abs(out['image_u8'].astype(int) - ref['image_u8'].astype(int)).max()
    assert d <= 1, tag + ': image differs by %d LSB' % d                       # tolerance: +-1 LSB per channel


@pytest.mark.parametrize("name", list(CONFIGS))
def test_config_matches_hostemu_and_oracle_windows(name, tmp_path, built):
    H, W, N, cam, rs, windows = CONFIGS[name]
    sc = h.Scene(tmp_path, H, W, N, cam=cam, render_scale=rs, seed0=4000)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0)
    rh = h.hb.RainHip(0)
    try:
        rh.set_streak_db(sc.db.streaks_light)
        rh.set_camera(sc.cam)
        fr = dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops)
        out = rh.render_frames([fr])[0]
        emu = h.emu_render(sc, bg, bg, env, drops)
        _assert_parity(out, emu, name + ' vs hostemu (%d drops)' % len(drops))
        # the product's default colour arithmetic (float, when no float64 composite is requested): same mask bits, image <= 1 LSB
        _assert_parity(rh.render_frames([fr], want_composite=False)[0], emu, name + ' (float colours) vs hostemu')
        assert (out['status'] == 0).sum() > 0.9 * len(drops) and out['mask'].max() > 0
        for a, b in windows:
            b = min(b, len(drops))
            assert b - a >= 100, 'window too small for %s: %d drops in the frame' % (name, len(drops))
            win = rh.render_frames([dict(fr, drops=drops[a:b])])[0]
            ref = h.oracle_render(sc, 0, bg, bg, env, faithful=True, first_drop=a, max_drops=b)
            _assert_parity(win, ref, '%s vs oracle, drops [%d, %d)' % (name, a, b))
    finally:
        rh.close()


def test_kitti100_whole_frame_against_the_numpy_oracle(tmp_path, built):
    """BASELINE.json configs[2], the headline workload: ALL streaks of a 1242x375 / 100 mm/hr frame (about 7200 after the
    frame filter) through the numpy oracle in its op-for-op mode -- about a minute of one host core -- against the kernels,
    both colour arithmetics: statuses equal, rainy_mask bit-exact (float64 and int32), rainy_image within 1 LSB."""
    H, W, N, cam, rs, _ = CONFIGS['kitti_100']
    sc = h.Scene(tmp_path, H, W, N, cam=cam, render_scale=rs, seed0=4000)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0)
    rh = h.hb.RainHip(0)
    try:
        rh.set_streak_db(sc.db.streaks_light)
        rh.set_camera(sc.cam)
        fr = dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops)
        out64 = rh.render_frames([fr])[0]
        out32 = rh.render_frames([fr], want_composite=False)[0]
    finally:
        rh.close()
    ref = h.oracle_render(sc, 0, bg, bg, env, faithful=True)
    assert len(drops) > 7000 and len(ref['status']) == len(drops)
    _assert_parity(out64, ref, 'kitti_100 whole frame (float64 colours) vs oracle')
    _assert_parity(out32, ref, 'kitti_100 whole frame (float colours) vs oracle')
    assert np.abs(out64['rainy_bg'] - ref['rainy_bg']).max() < 1e-9
