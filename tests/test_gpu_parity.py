"""GPU tier: the HIP path, called through the C ABI (ctypes), against the numpy oracle on
the same seeded inputs.  Bar: rainy_mask bit-exact (float64 accumulator and the int32
export), rainy_image within 1 LSB per channel, identical per-drop skip status."""
import numpy as np
import pytest

import helpers as h

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rh(built):
    ctx = h.hb.RainHip(0)
    yield ctx
    ctx.close()


def _render(rh, sc, i, bg, rainy, env, drops, opacity=1.0):
    rh.set_streak_db(sc.db.streaks_light)
    rh.set_camera(sc.cam)
    return rh.render_frames([dict(bg=bg, rainy_bg=rainy, env_xyY=env, omega=sc.omega, drops=drops,
                                  opacity_attenuation=opacity)])[0]


def _check(out, ref, tag=''):
    assert np.array_equal(out['status'], ref['status']), tag + ' status'
    assert np.array_equal(out['mask'], ref['mask']), tag + ' mask f64: max|d|=%g' % np.abs(out['mask'] - ref['mask']).max()
    assert np.array_equal(out['mask_i32'], ref['mask_i32']), tag + ' mask i32'
    d = np.abs(out['image_u8'].astype(int) - ref['image_u8'].astype(int)).max()
    assert d <= 1, tag + ' image differs by %d LSB' % d          # tolerance: +-1 LSB per channel
    assert np.abs(out['rainy_bg'] - ref['rainy_bg']).max() < 1e-9, tag + ' composite'


@pytest.mark.parametrize("H,W,N,seed,noise", [(96, 160, 150, 10, 0.0), (128, 256, 200, 20, 3.0), (64, 64, 80, 30, 0.0),
                                              (256, 256, 512, 50, 0.0)])
def test_gpu_matches_oracle(rh, tmp_path, H, W, N, seed, noise):
    sc = h.Scene(tmp_path, H, W, N, seed0=seed)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0, noise_std=noise, noise_scale=1.0)
    out = _render(rh, sc, 0, bg, bg, env, drops)
    ref = h.oracle_render(sc, 0, bg, bg, env, faithful=(N <= 200), noise_std=noise, noise_scale=1.0)
    _check(out, ref, '%dx%d' % (W, H))


def test_gpu_matches_hostemu_kitti_shape(rh, tmp_path):
    """KITTI-shaped frame, 25 mm/hr drop count: GPU vs the g++ build of the same arithmetic
    (bit-exact everywhere except the colour sums' order).  The numpy oracle on this configuration:
    tests/test_gpu_configs.py::kitti_25 (500-drop window)."""
    sc = h.Scene(tmp_path, 375, 1242, 2048, seed0=77)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0)
    out = _render(rh, sc, 0, bg, bg, env, drops)
    emu = h.emu_render(sc, bg, bg, env, drops)
    _check(out, emu, 'kitti-vs-hostemu')


def test_gpu_batch_of_frames_and_empty_frame(rh, tmp_path):
    sc = h.Scene(tmp_path, 96, 160, 100, n_frames=3, seed0=5)
    rh.set_streak_db(sc.db.streaks_light)
    rh.set_camera(sc.cam)
    frames, refs = [], []
    for i in range(3):
        bg, env = sc.frame_inputs(i)
        drops = sc.product_drops(i) if i != 1 else np.zeros(0, h.hb.DROP_DTYPE)
        frames.append(dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops))
        refs.append(h.emu_render(sc, bg, bg, env, drops))
    outs = rh.render_frames(frames)
    for o, r in zip(outs, refs):
        _check(o, r, 'batch')
    assert outs[1]['mask'].max() == 0


def test_gpu_fogged_background_and_opacity(rh, tmp_path):
    sc = h.Scene(tmp_path, 96, 160, 120, seed0=9)
    bg, env = sc.frame_inputs(0)
    rainy = np.clip(bg * 0.8 + 0.1, 0, 1)
    drops = sc.product_drops(0)
    out = _render(rh, sc, 0, bg, rainy, env, drops, opacity=0.7)
    textures, ratio = sc.oracle_db()
    ref = h.orc.render_frame(bg, rainy, env, sc.omega, sc.oracle_streaks(0), textures, ratio, sc.ocam, frame_seed=0,
                             opacity_attenuation=0.7, faithful=True)
    _check(out, ref, 'fog')


def test_gpu_white_strategy(rh, tmp_path):
    """rendering_strategy='white' (bad_weather.py:349-353): gray streaks, no defocus, numpy-slice placement
    (negative origins wrap), nothing skipped."""
    sc = h.Scene(tmp_path, 96, 160, 250, seed0=41, far_fraction=0.1)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0)
    rh.set_streak_db(sc.db.streaks_light)
    rh.set_camera(sc.cam)
    out = rh.render_frames([dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=drops, strategy=1)])[0]
    ref = h.oracle_render(sc, 0, bg, bg, env, strategy='white')
    assert not out['status'].any()
    assert np.array_equal(out['mask'], ref['mask'])
    assert np.array_equal(out['mask_i32'], ref['mask_i32'])
    assert np.array_equal(out['rainy_bg'], ref['rainy_bg'])           # no colour constant involved: bit-exact composite
    assert np.abs(out['image_u8'].astype(int) - ref['image_u8'].astype(int)).max() <= 1
    assert (drops['x0'] < 0).any() or (drops['y0'] < 0).any()          # the wrap-around path is exercised


def test_gpu_int32_mask_equals_untouched_reference(rh, tmp_path):
    """The 620-streak scene of tests/golden/make_golden.py section 8c: the REFERENCE's own add_drop_to_image with the
    real scipy.ndimage.gaussian_filter.  The kernels' deterministic exp moves blurred alphas by <= 4 ulp; the
    graded quantities must not notice: int32 mask identical, uint8 image within 1 LSB."""
    import os
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_vectors.npz'))
    BH, BW, BN, BSEED = (int(v) for v in G['big_scene'])
    sc = h.Scene(tmp_path, BH, BW, BN, seed0=BSEED, far_fraction=0.05)
    bg, env = sc.frame_inputs(0)
    out = _render(rh, sc, 0, bg, bg, env, sc.product_drops(0))
    assert np.array_equal((out['status'] != 0).astype(int), G['big_scipy_skipped'])
    assert np.array_equal(out['mask_i32'], G['big_scipy_mask_i32'])
    assert np.abs(out['image_u8'].astype(int) - G['big_scipy_image_u8'].astype(int)).max() <= 1     # +-1 LSB
