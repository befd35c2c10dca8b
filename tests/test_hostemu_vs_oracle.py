"""CPU tier: the kernel arithmetic (rr_device.h compiled with g++, tests/hostemu) against
the numpy oracle.  This is what lets arithmetic bugs surface without a GPU; the GPU tier
repeats the same comparisons through the real kernels."""
import numpy as np
import pytest

import helpers as h


@pytest.mark.parametrize("H,W,N,seed,noise", [(96, 160, 150, 10, 0.0), (128, 256, 200, 20, 3.0), (64, 64, 80, 30, 0.0)])
def test_hostemu_matches_oracle(tmp_path, H, W, N, seed, noise):
    sc = h.Scene(tmp_path, H, W, N, seed0=seed)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0, noise_std=noise, noise_scale=1.0)
    emu = h.emu_render(sc, bg, bg, env, drops)
    ref = h.oracle_render(sc, 0, bg, bg, env, faithful=True, noise_std=noise, noise_scale=1.0)
    assert np.array_equal(emu['status'], ref['status'])
    assert np.array_equal(emu['mask'], ref['mask'])
    assert np.array_equal(emu['mask_i32'], ref['mask_i32'])
    assert np.abs(emu['image_u8'].astype(int) - ref['image_u8'].astype(int)).max() <= 1
    assert np.abs(emu['rainy_bg'] - ref['rainy_bg']).max() < 1e-12


def test_hostemu_forced_paths(tmp_path):
    """The hand-built edge-case scene of the GPU tier (integer-ratio and bilinear INTER_AREA
    modes, border crops, skipped and heavily defocused drops), kernel arithmetic vs oracle."""
    import ctypes
    import test_gpu_edge_cases as e
    sc = h.Scene(tmp_path, e.H, e.W, 0, frames=e._frames())
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0)
    emu = h.emu_render(sc, bg, bg, env, drops)
    ref = h.oracle_render(sc, 0, bg, bg, env, faithful=False)
    assert np.array_equal(emu['status'], ref['status'])
    assert np.array_equal(emu['mask'], ref['mask'])
    assert np.abs(emu['image_u8'].astype(int) - ref['image_u8'].astype(int)).max() <= 1
    # all three resize modes and both drop kinds are present in this scene
    lib = h.hostemu()
    n, psz = len(drops), lib.emu_sizeof_plan()
    plans = np.zeros(n * psz, np.uint8)
    poly = np.zeros(n * 72, np.int32)
    npts = np.zeros(n, np.int32)
    sizes = np.zeros(n, np.int64)
    texels, hs, ws, offs = h.hb.pack_streak_db(sc.db.streaks_light)
    lib.emu_plan(h._p(drops), n, ctypes.byref(sc.cam), e.H, e.W, sc.He, sc.We, h._p(hs), h._p(ws), ctypes.c_double(1.0),
                 h._p(plans), h._p(poly), h._p(npts), h._p(sizes))
    ints = plans.reshape(n, psz)[:, :24 * 4].copy().view(np.int32).reshape(n, 24)
    kind, rs_mode = ints[:, 1], ints[:, 21]
    assert set(kind) == {0, 1}
    assert set(rs_mode[kind == 1]) == {0, 1, 2}
    assert (npts == 24).any() or (npts == 20).all()


@pytest.mark.parametrize("name,window", [('cityscapes_half', (200, 300)), ('cityscapes_full', (1000, 1060)),
                                         ('nuscenes_100', (3000, 3080)), ('kitti_100', (4000, 4150))])
def test_hostemu_matches_oracle_on_baseline_configs(tmp_path, name, window):
    """Windows of the BASELINE.json configurations (Cityscapes 5 ms / render_scale 2, nuScenes f/1.8 5.5 mm,
    KITTI 100 mm/hr) at their full frame and environment-map sizes: kernel arithmetic vs the op-for-op oracle.
    The GPU tier (tests/test_gpu_configs.py) repeats this through the kernels with larger windows."""
    import test_gpu_configs as cfg
    H, W, N, cam, rs, _ = cfg.CONFIGS[name]
    sc = h.Scene(tmp_path, H, W, N, cam=cam, render_scale=rs, seed0=4000)
    bg, env = sc.frame_inputs(0)
    a, b = window
    drops = sc.product_drops(0)[a:b]
    emu = h.emu_render(sc, bg, bg, env, drops)
    ref = h.oracle_render(sc, 0, bg, bg, env, faithful=True, first_drop=a, max_drops=b)
    assert np.array_equal(emu['status'], ref['status'])
    assert np.array_equal(emu['mask'], ref['mask'])
    assert np.array_equal(emu['mask_i32'], ref['mask_i32'])
    assert np.abs(emu['image_u8'].astype(int) - ref['image_u8'].astype(int)).max() <= 1
    assert (emu['status'] == 0).sum() > 0.8 * len(drops)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_depth_occlusion_hostemu_matches_oracle(tmp_path, dtype):
    """The depth-occlusion OPTION (RR_OPT_DEPTH_OCCLUSION; default off, not part of the reference's output) has two
    independent statements outside the library -- oracle/render.py (_visible, numpy) and tests/hostemu (C++ loops):
    they must agree bit for bit before the GPU tier compares the HIP option with both."""
    import test_gpu_depth_occlusion as tdo
    H, W = 128, 256
    sc = h.Scene(tmp_path, H, W, 400, seed0=21)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0)
    depth = tdo._scene_depth(H, W, np.abs(drops['wps'][:, 2]), dtype)
    emu = h.emu_render(sc, bg, bg, env, drops, depth=depth)
    ref = h.oracle_render(sc, 0, bg, bg, env, faithful=False, scene_depth=depth)
    base = h.emu_render(sc, bg, bg, env, drops)
    assert np.array_equal(emu['status'], ref['status'])
    assert np.array_equal(emu['mask'], ref['mask']) and np.array_equal(emu['mask_i32'], ref['mask_i32'])
    assert np.abs(emu['image_u8'].astype(int) - ref['image_u8'].astype(int)).max() <= 1
    assert 0.2 * base['mask'].sum() < emu['mask'].sum() < 0.9 * base['mask'].sum()


def test_whole_frame_kitti25_oracle_vs_hostemu(tmp_path):
    """EVERY drop of a BASELINE configs[1] frame (KITTI 1242x375, 25 mm/hr) through the numpy oracle in its op-for-op mode
    (per-drop masked reduction over the whole environment map, like the reference) against the g++ build of the kernel
    arithmetic: statuses, float64 mask, int32 mask bit for bit; image within 1 LSB.  (The windows elsewhere cover parts of a
    frame; the GPU tier compares the kernels with hostemu at full size, and with the whole-frame oracle at 100 mm/hr.)"""
    import test_gpu_configs as cfg
    H, W, N, cam, rs, _ = cfg.CONFIGS['kitti_25']
    sc = h.Scene(tmp_path, H, W, N, cam=cam, render_scale=rs, seed0=4000)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0)
    emu = h.emu_render(sc, bg, bg, env, drops)
    ref = h.oracle_render(sc, 0, bg, bg, env, faithful=True)
    assert len(drops) > 1500
    assert np.array_equal(emu['status'], ref['status'])
    assert np.array_equal(emu['mask'], ref['mask']) and np.array_equal(emu['mask_i32'], ref['mask_i32'])
    assert np.abs(emu['image_u8'].astype(int) - ref['image_u8'].astype(int)).max() <= 1


def test_whole_frame_cityscapes_default_oracle_vs_hostemu(tmp_path):
    """EVERY drop of a frame of BASELINE configs[3] as the Cityscapes plug-in renders it by default (render_scale 2:
    1024x512, 50 mm/hr, its own camera) through the op-for-op numpy oracle against the g++ build of the kernel arithmetic --
    the second configuration checked at whole-frame granularity (the GPU tier compares the kernels with hostemu at this
    size and at 2048x1024)."""
    import test_gpu_configs as cfg
    H, W, N, cam, rs, _ = cfg.CONFIGS['cityscapes_half']
    sc = h.Scene(tmp_path, H, W, N, cam=cam, render_scale=rs, seed0=4100)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0)
    emu = h.emu_render(sc, bg, bg, env, drops)
    ref = h.oracle_render(sc, 0, bg, bg, env, faithful=True)
    assert len(drops) > 1000
    assert np.array_equal(emu['status'], ref['status'])
    assert np.array_equal(emu['mask'], ref['mask']) and np.array_equal(emu['mask_i32'], ref['mask_i32'])
    assert np.abs(emu['image_u8'].astype(int) - ref['image_u8'].astype(int)).max() <= 1


def _window_job(args):
    """One window of a frame's drop list through both statements (a process of the pool below)."""
    import tempfile
    import test_gpu_configs as cfg
    name, a, b = args
    H, W, N, cam, rs, _ = cfg.CONFIGS[name]
    with tempfile.TemporaryDirectory() as tmp:
        sc = h.Scene(tmp, H, W, N, cam=cam, render_scale=rs, seed0=4000)
        bg, env = sc.frame_inputs(0)
        drops = sc.product_drops(0)[a:b]
        emu = h.emu_render(sc, bg, bg, env, drops)
        ref = h.oracle_render(sc, 0, bg, bg, env, faithful=True, first_drop=a, max_drops=b)
    ok = (np.array_equal(emu['status'], ref['status']) and np.array_equal(emu['mask'], ref['mask']) and
          np.array_equal(emu['mask_i32'], ref['mask_i32']))
    return a, b, bool(ok), int(np.abs(emu['image_u8'].astype(int) - ref['image_u8'].astype(int)).max()), int((emu['status'] == 0).sum())


@pytest.mark.parametrize("name,every", [('cityscapes_full', 1), ('nuscenes_200', 4)])
def test_every_drop_of_the_large_configs_oracle_vs_hostemu(name, every):
    """BASELINE configs[3] at full size (2048x1024, every one of its ~3600 drops) and configs[4] at 200 mm/hr (1600x900,
    ~14 000 drops: every fourth window of 200) through the numpy oracle in its op-for-op mode against the g++ build of the
    kernel arithmetic, window by window on a pool of processes (the masked reduction over a 3 M-texel environment map costs
    45 ms per drop on one core): status, float64 and int32 mask bit for bit, image within 1 LSB in every window.  (A window
    is composited onto the plain background: what the windows do not cover is the accumulation ACROSS windows, which the
    whole-frame tests above and the GPU tier's full-size comparison with hostemu do.)"""
    import multiprocessing as mp
    import os
    import tempfile
    import test_gpu_configs as cfg
    H, W, N, cam, rs, _ = cfg.CONFIGS[name]
    with tempfile.TemporaryDirectory() as tmp:
        n = len(h.Scene(tmp, H, W, N, cam=cam, render_scale=rs, seed0=4000).product_drops(0))
    step = 200
    jobs = [(name, a, min(a + step, n)) for k, a in enumerate(range(0, n, step)) if k % every == 0]
    with mp.get_context('fork').Pool(min(len(jobs), max(1, min(8, os.cpu_count() or 1)))) as pool:
        res = pool.map(_window_job, jobs)
    assert sum(b - a for a, b, *_ in res) >= (n if every == 1 else n // (every + 1))
    for a, b, ok, d, kept in res:
        assert ok, '%s drops [%d, %d): status / mask differ' % (name, a, b)
        assert d <= 1, '%s drops [%d, %d): image differs by %d LSB' % (name, a, b, d)
    assert sum(r[4] for r in res) > 0.8 * sum(b - a for a, b, *_ in res)


def sliver_scene(tmp_path, monkeypatch, fov_deg=0.6):
    """A camera whose field-of-view cone is a fraction of a degree: the truncated polygons cover a texel or two -- some
    have three vertices that are not collinear, most do not.  Oracle and product get the same cone angle."""
    sc = h.Scene(tmp_path, 96, 160, 150, seed0=77)
    cs = sc.cam_settings
    sc.cam = h.hb.make_camera(cs['focal_mm'] / 1000., cs['f_number'], cs['exposure_ms'], fov=fov_deg)
    monkeypatch.setattr(h.orc, 'FOV_DEG', fov_deg)
    return sc


def test_collinear_polygons_are_skipped(tmp_path, monkeypatch):
    """pyclipper's AddPath (bad_weather.py:368) raises for a path without three non-collinear vertices: such a drop is
    skipped by the reference's caller (generator.py:180-189).  Oracle (cvlike.polygon_all_collinear) and kernel arithmetic
    (rr_device.h poly_all_collinear; the float polygon hands every sliver to float64) give the same statuses and mask."""
    sc = sliver_scene(tmp_path, monkeypatch)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0)
    emu = h.emu_render(sc, bg, bg, env, drops)
    ref = h.oracle_render(sc, 0, bg, bg, env, faithful=True)
    assert np.array_equal(emu['status'], ref['status'])
    assert np.array_equal(emu['mask'], ref['mask'])
    assert np.abs(emu['image_u8'].astype(int) - ref['image_u8'].astype(int)).max() <= 1
    st = ref['status']
    assert (st == h.orc.ST_FOV_FAIL).sum() >= 10 and (st == 0).sum() >= 10, np.bincount(st)
