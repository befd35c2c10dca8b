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


def test_main_device_particles_on_a_nuscenes_tree(tmp_path, built, monkeypatch):
    """BASELINE.json configs[4] from the command line, on the GPU: `main.py --dataset nuscenes --device_particles` on a
    nuScenes-shaped tree (1600x900, f/1.8) -- no particle file is read or written, the drop tables are born on the device --
    against rr_pipeline_submit fed the very rr_sim_frame records the driver must have sent (tools/particles.sim_frames,
    simulated frame f % n_sim with the draws of frame f: reference generator.py:304-321): the PNG files decode to the same
    pixels.  Reference role: main.py:187-220 (particles resolution / auto-simulation)."""
    tmp = str(tmp_path)
    H, W, n = 900, 1600, 3
    scene = os.path.join(tmp, 'source', 'nuscenes', 'scene-0001')
    os.makedirs(os.path.join(scene, 'rgb'))
    os.makedirs(os.path.join(scene, 'depth'))
    for i in range(n):
        bgr = h.synthetic.make_frame(40 + i, H, W)
        Image.fromarray((bgr[..., ::-1] * 255).astype(np.uint8)).save(os.path.join(scene, 'rgb', '%06d.png' % i))
        ramp = np.linspace(80.0, 2.0, H)[:, None] * np.ones((1, W))
        Image.fromarray(np.round(ramp * 256).astype(np.uint16)).save(os.path.join(scene, 'depth', '%06d.png' % i))
    tex_dir, norm = h.synthetic.write_streak_db(os.path.join(tmp, 'rainstreakdb'))
    monkeypatch.setenv('RAIN_BATCH', '2')                              # two batches, the second ragged
    main = importlib.import_module('rain-rendering_amd.main')
    src = os.path.join(tmp, 'source')
    gen = main.main(['--dataset', 'nuscenes', '-k', src, '-d', src, '-r', os.path.join(tmp, 'particles'), '-sd', os.path.join(tmp, 'rainstreakdb'),
                     '-i', '5', '--output', os.path.join(tmp, 'out'), '--noverbose', '--device_particles'])
    assert not os.path.exists(os.path.join(tmp, 'particles'))          # no particle file appeared
    assert gen.timing[0].get('route') == 'native' and len(gen.stats) == n and all(s['drops'] > 50 for s in gen.stats)
    out_dir = os.path.join(tmp, 'out', 'nuscenes', 'scene-0001', 'rain', '5mm')

    # the same frames through the library, fed the records the driver sends
    particles = importlib.import_module('rain-rendering_amd.tools.particles')
    dbmod = importlib.import_module('rain-rendering_amd.common.db')
    imgops = importlib.import_module('rain-rendering_amd.common.imgops')
    fogmod = importlib.import_module('rain-rendering_amd.common.add_attenuation')
    envmod = importlib.import_module('rain-rendering_amd.common.envmap')
    st = dbmod.settings('nuscenes')
    opts = dbmod.sim('nuscenes', 'scene-0001', os.path.join(tmp, 'particles', 'nuscenes'))['options']
    n_sim = particles.n_sim_frames(opts)
    sims, dgrid, cdf = particles.sim_frames(opts, 5, n_sim, render_scale=1, seed=0)
    f_idx = [int(v) for v in np.linspace(0, n_sim, n, endpoint=False, dtype=int)]      # generator.py:304-312
    db = h.bw.DBManager(streaks_path=tex_dir, norm_coeff_path=norm)
    db.load_streak_database()
    focal = st['cam_focal'] / 1000.
    consts = fogmod.FogRain(rain_intensity=5, focal=focal, f_number=st['cam_f_number'], angle=90, exposure=st['cam_exposure'],
                            camera_gain=st['cam_gain']).constants()
    rh = h.hb.RainHip(0)
    try:
        rh.set_streak_db(db.streaks_light)
        rh.set_camera(h.hb.make_camera(focal, st['cam_f_number'], st['cam_exposure']))
        rh.set_prepass_kernels(imgops.gaussian_kernel(25, 25), imgops.gaussian_kernel(15, 0))
        rh.set_colormap(imgops.viridis_lut())
        rh.set_particle_tables(dgrid, cdf)
        we = rh.set_envmap_geometry(H, W, *envmod.EnvironmentMapGenerator(focal, W, H).device_tables(H, W))
        omega = h.solid_angle.get_solid_angles(np.empty((H, we, 0)))
        frames, outs = [], []
        for i in range(n):
            rec = sims[f_idx[i] % n_sim:f_idx[i] % n_sim + 1].copy()
            rec['draw_seed'] = f_idx[i]
            bg8 = imgops.imread_bgr(os.path.join(scene, 'rgb', '%06d.png' % i))
            depth = imgops.imread_unchanged(os.path.join(scene, 'depth', '%06d.png' % i)).astype(np.float32) / 256.
            cap = int(sims['n_particles'].max())
            frames.append(dict(bg_u8=np.ascontiguousarray(bg8), depth=np.ascontiguousarray(depth), fog=consts, omega=omega, sim=rec, drops_cap=cap))
            outs.append(dict(image_u8=np.zeros((H, W, 3), np.uint8), mask=np.zeros((H, W)), status=np.zeros(cap, np.int32),
                             n_drops=np.zeros(1, np.int32)))
        rh.pipeline_submit(0, frames, outs)
        while not rh.pipeline_wait(0):
            rh.pipeline_submit(0, frames, outs)
    finally:
        rh.close()
    by_file = {os.path.basename(s['file']): s for s in gen.stats}
    for i in range(n):
        name = '%06d.png' % i
        got = np.array(Image.open(os.path.join(out_dir, 'rainy_image', name)))
        assert got.shape == (H, W, 4) and np.array_equal(got[..., :3], outs[i]['image_u8']), name
        ref_mask = os.path.join(tmp, 'ref_mask.png')
        imgops.imsave_scalar(ref_mask, outs[i]['mask'])
        assert np.array_equal(np.array(Image.open(os.path.join(out_dir, 'rain_mask', name))), np.array(Image.open(ref_mask))), name
        assert by_file[name]['drops'] == int(outs[i]['n_drops'][0]) and outs[i]['mask'].max() > 0


def test_rccl_broadcast_of_the_streak_database_world1(tmp_path, built):
    """The N>1 start-up on the one GPU of the test tier: init_process_group('nccl', device_id=...) with a single rank, then
    sharding.load_and_broadcast_streak_db's collective route -- RCCL broadcast of the header and of the packed database as
    DEVICE tensors, rr_set_streak_db_device on the received buffer -- and a frame rendered from it, equal to a frame rendered
    from a plainly uploaded database.  (The 8-GPU scaling run is the driver's; this is the code it executes first.)  In a
    process of its own that imports torch BEFORE the library is loaded, the order bench.py and main.py use under a launcher
    (this pytest process has librainhip.so's HIP runtime loaded already).  Reference role: main_threaded.py:98-200."""
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'nccl_world1_worker.py')
    r = subprocess.run([sys.executable, worker, str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))
    assert r.returncode == 0 and b'NCCL-WORLD1-OK' in r.stdout, (r.stdout.decode()[-2000:], r.stderr.decode()[-4000:])


def test_two_ranks_on_the_one_gpu_write_the_single_rank_files(tmp_path, built):
    """The real driver under N > 1 on the hardware there is (the role of the reference's main_threaded.py:98-200):
    `python -m torch.distributed.run --nproc-per-node 2 rain-rendering_amd/main.py ...`, both ranks on GPU 0 (RAIN_DEVICE=0) with
    gloo for the process group (RCCL refuses two ranks on one device; the one-rank RCCL path is the test above).  Rank 0 names
    the output folder and the work list and loads the streak database, the broadcast hands them over (device tensors), each
    rank renders its round-robin share through its own context on the shared GPU: every frame is written exactly once into ONE
    folder and every file is byte for byte the file of the single-rank run.  Then bench.py's strong-scaling launch line with
    two ranks on the same GPU (a smoke of the line the driver uses for SCALE_rNN.json; not a scaling number)."""
    import json
    import socket
    import subprocess
    import sys
    tmp = str(tmp_path)
    n = 12
    src, xml = _make_dataset(tmp, n_frames=n)
    main = importlib.import_module('rain-rendering_amd.main')
    common = ['--dataset', 'kitti', '-k', src, '-d', src, '-r', os.path.join(tmp, 'particles'), '-sd', os.path.join(tmp, 'rainstreakdb'),
              '-i', '5', '--noverbose']
    main.main(common + ['--output', os.path.join(tmp, 'out1')])

    def free_port():
        with socket.socket() as s_:
            s_.bind(('127.0.0.1', 0))
            return s_.getsockname()[1]
    env = dict(os.environ, RAIN_DEVICE='0', RAIN_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', RAIN_BATCH='4')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(free_port()), os.path.join(h.ROOT, 'rain-rendering_amd', 'main.py')] + common +
                       ['--output', os.path.join(tmp, 'out2'), '--conflict_strategy', 'rename_folder'],
                       env=env, cwd=h.ROOT, capture_output=True, timeout=600)
    assert r.returncode == 0, (r.stdout.decode()[-3000:], r.stderr.decode()[-3000:])
    sub = os.path.join('kitti', 'data_object', 'training', 'rain', '5mm')
    for kind in ('rainy_image', 'rain_mask'):
        a, b = os.path.join(tmp, 'out1', sub, kind), os.path.join(tmp, 'out2', sub, kind)
        assert sorted(os.listdir(a)) == sorted(os.listdir(b)) == ['%06d.png' % i for i in range(n)]
        for f in os.listdir(a):
            assert open(os.path.join(a, f), 'rb').read() == open(os.path.join(b, f), 'rb').read(), (kind, f)
    assert sorted(os.listdir(os.path.join(tmp, 'out2', 'kitti', 'data_object', 'training', 'rain'))) == ['5mm']     # ONE folder
    # bench.py, two ranks, one 64-frame sequence (strong scaling)
    benv = dict(os.environ, RAIN_BENCH_DEVICE='0', RAIN_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    b = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(free_port()), os.path.join(h.ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '32',
                        '--total-frames', '64', '--no-cpu-baseline', '--no-prepass', '--no-variants', '--no-traffic', '--no-driver'],
                       env=benv, cwd=h.ROOT, capture_output=True, timeout=900)
    assert b.returncode == 0, (b.stdout.decode()[-2000:], b.stderr.decode()[-3000:])
    lines = [l for l in b.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, lines                                              # rank 0 alone prints
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong' and line['value'] > 0 and line['steps'] == 2
    assert line['config']['frames_per_step'] == 64 and 'dp2' in line['config']['parallelism']
