"""GPU tier: the main.py-compatible driver end to end on a synthetic on-disk dataset (the
reference's input trees and output tree), and its frames against the oracle."""
import importlib
import os

import numpy as np
import pytest
from PIL import Image

import helpers as h
from oracle import render as orc

pytestmark = pytest.mark.gpu


def _make_dataset(tmp, n_frames=3, H=96, W=160, rate=5):
    src = os.path.join(tmp, 'source')
    h.synthetic.write_dataset(src, 'kitti', os.path.join('data_object', 'training'), n_frames, H, W)
    tex_dir, norm = h.synthetic.write_streak_db(os.path.join(tmp, 'rainstreakdb'))
    frames = h.synthetic.simulate_particles(2, 150, W, H)
    xml = os.path.join(tmp, 'particles', 'kitti', 'data_object', 'rain', '%dmm' % rate, 'sim_camera0.xml')
    h.synthetic.write_particles_xml(xml, frames)
    return src, xml


def test_main_cli_end_to_end(tmp_path, built):
    tmp = str(tmp_path)
    src, xml = _make_dataset(tmp)
    main = importlib.import_module('rain-rendering_amd.main')
    argv = ['--dataset', 'kitti', '-k', src, '-d', src, '-r', os.path.join(tmp, 'particles'), '-sd',
            os.path.join(tmp, 'rainstreakdb'), '-i', '5', '--output', os.path.join(tmp, 'out'), '--noverbose', '--save_envmap']
    gen = main.main(argv)
    out_dir = os.path.join(tmp, 'out', 'kitti', 'data_object', 'training', 'rain', '5mm')
    for i in range(3):
        p = os.path.join(out_dir, 'rainy_image', '%06d.png' % i)
        m = os.path.join(out_dir, 'rain_mask', '%06d.png' % i)
        assert os.path.exists(p) and os.path.exists(m)
        assert np.array(Image.open(p)).shape == (96, 160, 4)
    assert os.path.exists(os.path.join(tmp, 'out', 'kitti', 'data_object', 'training', 'envmap', '000000.png'))
    assert len(gen.stats) == 3 and all(s['drops'] > 50 for s in gen.stats)

    # frame 1 against the oracle pipeline (numpy pre-pass + numpy hot path)
    from oracle import prepass as opre
    imgops = importlib.import_module('rain-rendering_amd.common.imgops')
    i = 1
    img_dir = os.path.join(src, 'kitti', 'data_object', 'training', 'image_2')
    bg = imgops.imread_bgr(os.path.join(img_dir, '%06d.png' % i)) / 255.0
    depth = imgops.imread_unchanged(os.path.join(img_dir, 'depth', '%06d.png' % i)).astype(np.float32) / 256.
    rainy = opre.fog_rain_layer(bg, depth, 5, 6.0, 2, 20)
    env_bgr = opre.generate_env_map(rainy, 0.006)
    env = opre.env_to_xyY(env_bgr)
    omega = h.solid_angle.get_solid_angles(env_bgr)
    sim = orc.load_streaks_from_xml(xml, 1, [160, 96])
    fr = list(sim.values())[i % 2]
    streaks = list(orc.streak_filter(fr.streaks, 160, 96).values())
    textures, ratio = orc.load_streak_database(os.path.join(tmp, 'rainstreakdb', 'env_light_database', 'size32'),
                                               os.path.join(tmp, 'rainstreakdb', 'env_light_database', 'txt', 'normalized_env_max.txt'))
    ref = orc.render_frame(bg, rainy, env, omega, streaks, textures, ratio,
                           dict(focal_m=0.006, f_number=6.0, exposure_ms=2), frame_seed=i, faithful=True)
    got = np.array(Image.open(os.path.join(out_dir, 'rainy_image', '%06d.png' % i)))[..., :3]
    assert np.abs(got.astype(int) - ref['image_u8'].astype(int)).max() <= 1        # +-1 LSB

    # conflict_strategy skip: a second run renders nothing new
    argv2 = argv[:-1] + ['--conflict_strategy', 'skip']
    gen2 = main.main(argv2)
    assert len(gen2.stats) == 0


def test_compute_drop_seam_equals_batched_call(tmp_path, built):
    """The reference-shaped single-drop seam Generator.compute_drop (generator.py:119-191), looped over a
    frame's streaks in order, reproduces the batched rr_render_frames result bit for bit."""
    import types
    gen_mod = importlib.import_module('rain-rendering_amd.common.generator')
    sc = h.Scene(tmp_path, 64, 96, 60, seed0=23)
    bg, env = sc.frame_inputs(0)
    rh = h.hb.RainHip(0)
    rh.set_streak_db(sc.db.streaks_light)
    rh.set_camera(sc.cam)
    batched = rh.render_frames([dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=sc.product_drops(0))])[0]
    # fresh loaders (product_drops consumed the RNG and may have mutated the table)
    sc2 = h.Scene(tmp_path / 'b', 64, 96, 60, seed0=23)
    g = gen_mod.Generator.__new__(gen_mod.Generator)
    g.db, g._hip, g.noise_std, g.noise_scale, g.opacity_attenuation, g.rendering_strategy = sc2.db, rh, 0.0, 0.0, 1.0, None
    g.env_map_xyY, g.solid_angle_map = env, sc2.omega
    g.renderer = h.bw.RainRenderer(focal=sc2.ocam['focal_m'], f_number=sc2.ocam['f_number'], focus_plane=6, radius=10, fov=165)
    fr = list(sc2.db.streaks_simulator.values())[0]
    keep = h.hb.filter_streaks(fr.table, 96, 64)
    streaks = [fr.table.streak(int(i)) for i in keep]
    np.random.seed(0)
    rainy, mask, sat = bg.copy(), np.zeros((64, 96)), np.zeros((64, 96, 3))
    rain_layer = np.zeros((64, 96, 4))
    skipped = 0
    for s in streaks:                                  # the reference's loop body (generator.py:431-438)
        rainy, mask, sat, drop, blended, minC = g.compute_drop(bg, s, rainy, mask, sat)
        if blended is not None:
            assert drop.shape[:2] == blended.shape[:2] and drop.shape[2] == 4
            # the returned tile sits where the blend happened: its alpha is what the mask gained there
            ys, xs = int(minC[1]), int(minC[0])
            assert np.array_equal(blended, rainy[ys:ys + blended.shape[0], xs:xs + blended.shape[1]])
            rain_layer = g.renderer.make_rain_layer(drop, blended, rain_layer, mask, minC)
        skipped += blended is None
    assert np.array_equal(rain_layer[..., 3] > 0, mask > 0)
    assert skipped == int(np.count_nonzero(batched['status']))
    assert np.array_equal(mask, batched['mask'])
    assert np.array_equal(rainy, batched['rainy_bg'])
    rh.close()


def test_native_and_general_routes_write_the_same_files(tmp_path, built, monkeypatch):
    """The driver's batch-native route (rr_io_read_frames / rr_host_pack_frames / rr_io_write_frames: one library call
    per batch and stage) and its general route (per-frame Python on I/O threads) on the same dataset: byte-identical
    output files, frame by frame; several batches with a ragged last one."""
    tmp = str(tmp_path)
    src, xml = _make_dataset(tmp, n_frames=7)
    main = importlib.import_module('rain-rendering_amd.main')
    outs = {}
    for route, flag in (('native', '1'), ('general', '0')):
        monkeypatch.setenv('RAIN_NATIVE_IO', flag)
        monkeypatch.setenv('RAIN_BATCH', '3')
        argv = ['--dataset', 'kitti', '-k', src, '-d', src, '-r', os.path.join(tmp, 'particles'), '-sd',
                os.path.join(tmp, 'rainstreakdb'), '-i', '5', '--output', os.path.join(tmp, 'out_' + route), '--noverbose']
        gen = main.main(argv)
        assert (gen.timing[0].get('route') == 'native') == (route == 'native')
        assert len(gen.stats) == 7 and all(s['drops'] > 50 for s in gen.stats)
        outs[route] = os.path.join(tmp, 'out_' + route, 'kitti', 'data_object', 'training', 'rain', '5mm')
    for i in range(7):
        for kind in ('rainy_image', 'rain_mask'):
            a = open(os.path.join(outs['native'], kind, '%06d.png' % i), 'rb').read()
            b = open(os.path.join(outs['general'], kind, '%06d.png' % i), 'rb').read()
            assert a == b and len(a) > 500, (kind, i)
    mask = np.array(Image.open(os.path.join(outs['native'], 'rain_mask', '000003.png')))
    assert len(np.unique(mask.reshape(-1, 4), axis=0)) > 3                  # streaks were rendered
