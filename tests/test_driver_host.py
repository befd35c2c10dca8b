"""CPU tier: the driver's host pipeline (Generator._run_batches, both routes) around a stand-in for the GPU context that
records what reaches `rr_pipeline_submit` and delivers known scanlines: the frames, depth maps and drop tables the
library would get are the ones the per-frame loaders make, skipped frames leave no gap, every output file is written from
its own frame's scanlines, and the batch-native route feeds the library exactly what the general route does."""
import importlib
import os
import sys

import numpy as np
import pytest
from PIL import Image

import helpers as h

sys.path.insert(0, os.path.join(h.ROOT, 'scripts'))
import driver_host_only as dho                                     # noqa: E402  (the stand-in context lives with the script)

hb = h.hb
generator_mod = importlib.import_module('rain-rendering_amd.common.generator')
main_mod = importlib.import_module('rain-rendering_amd.main')
imgops = importlib.import_module('rain-rendering_amd.common.imgops')


def _frame_inputs(fr):
    """(image, depth) of a prepared frame in whichever form the driver hands them over: pixels, or -- the batch-native route --
    the files' filtered scanlines (un-filtered here by the host build of the device's rule)."""
    if fr.get('bg_png_rows') is not None:
        import test_pngrows_host as tr
        H, W = fr['shape']
        return tr._unfilter(np.ascontiguousarray(fr['bg_png_rows']), H, W, 3), tr._unfilter(np.ascontiguousarray(fr['depth_png_rows']), H, W, 2)
    return (fr['bg_u8'] if fr.get('bg_u8') is not None else fr['bg']), fr['depth']


class RecordingContext(dho.HostOnlyContext):
    """Keeps a copy of every submitted frame's inputs; "renders" scanlines that encode the frame's identity (the first
    pixel of its image), so that a file written from another frame's buffer would be noticed."""
    log = None

    def pipeline_submit_prepared(self, slot, prep, n=None):
        n = prep.n if n is None else n
        for k in range(n):
            fr = prep.frames[k]
            nd = prep.counts.get(k, len(fr['drops']))
            bg, depth = _frame_inputs(fr)
            RecordingContext.log.append(dict(bg=np.array(bg), depth=np.array(depth), drops=np.array(fr['drops'][:nd]),
                                             sim=None if fr.get('sim') is None else np.array(fr['sim'][0])))
        super().pipeline_submit_prepared(slot, prep, n)

    def pipeline_prepare(self, frames, outs):
        p = super().pipeline_prepare(frames, outs)
        p.counts = {}
        p.set_drop_count = lambda k, n_: p.counts.__setitem__(k, int(n_))
        return p

    def pipeline_wait(self, slot):
        prep, n = self.batches.pop(slot, (None, 0))
        if prep is None:
            return True
        for k in range(n):
            o, fr = prep.outs[k], prep.frames[k]
            bg = _frame_inputs(fr)[0]
            bg = bg if bg.dtype == np.uint8 else np.round(bg * 255).astype(np.uint8)
            H, W = bg.shape[:2]
            rows = np.zeros((H, 1 + 4 * W), np.uint8)               # filter type 0 rows: every pixel = the frame's first pixel
            rows[:, 1:] = np.tile(np.append(bg[0, 0], 255).astype(np.uint8), W)
            o['rainy_png'][...] = rows.ravel()
            rows[:, 1:] = np.tile(np.array([k % 251, bg[0, 0, 1], 7, 255], np.uint8), W)
            o['mask_png'][...] = rows.ravel()
            o['status'][...] = 0
            if o.get('n_drops') is not None:                       # (what the device-side generator would report)
                o['n_drops'][0] = 100 + int(fr['sim'][0]['draw_seed'])
        return True


def _dataset(tmp, n_frames, H=48, W=80):
    src = os.path.join(tmp, 'source')
    img_dir, dep_dir = h.synthetic.write_dataset(src, 'kitti', os.path.join('data_object', 'training'), n_frames, H, W)
    h.synthetic.write_streak_db(os.path.join(tmp, 'rainstreakdb'))
    frames = h.synthetic.simulate_particles(2, 200, W, H)
    xml = os.path.join(tmp, 'particles', 'kitti', 'data_object', 'rain', '5mm', 'sim_camera0.xml')
    h.synthetic.write_particles_xml(xml, frames)
    return src, img_dir, dep_dir


def _run(tmp, src, out_name, native, monkeypatch, batch=3):
    monkeypatch.setenv('RAIN_BATCH', str(batch))
    monkeypatch.setenv('RAIN_NATIVE_IO', '1' if native else '0')
    monkeypatch.setattr(hb, 'RainHip', RecordingContext)
    RecordingContext.log = []
    argv = ['--dataset', 'kitti', '-k', src, '-d', src, '-r', os.path.join(tmp, 'particles'), '-sd', os.path.join(tmp, 'rainstreakdb'),
            '-i', '5', '--output', os.path.join(tmp, out_name), '--noverbose']
    gen = main_mod.main(argv)
    return gen, RecordingContext.log, os.path.join(tmp, out_name, 'kitti', 'data_object', 'training', 'rain', '5mm')


def test_native_route_feeds_the_library_what_the_general_route_does(tmp_path, built, monkeypatch):
    tmp = str(tmp_path)
    n = 8
    src, img_dir, dep_dir = _dataset(tmp, n)
    # frame 2: a corrupt depth file -- both routes skip the frame (generator.py:360-363) and close the gap in its batch
    with open(os.path.join(dep_dir, '%06d.png' % 2), 'wb') as fh:
        fh.write(b'not a png')
    gen_n, log_n, out_n = _run(tmp, src, 'out_native', True, monkeypatch)
    gen_g, log_g, out_g = _run(tmp, src, 'out_general', False, monkeypatch)
    assert gen_n.timing[0].get('route') == 'native' and gen_g.timing[0].get('route') != 'native'
    assert len(gen_n.stats) == len(gen_g.stats) == n - 1 == len(log_n) == len(log_g)

    def by_identity(log):
        return {fr['bg'].tobytes(): fr for fr in log}
    a, b = by_identity(log_n), by_identity(log_g)
    assert a.keys() == b.keys() and len(a) == n - 1                 # the same frames (batches may order them differently)
    for key in a:
        # (the batch-native route hands over the depth file's uint16 samples, RR_DEPTH_U16: metres = sample / 256 on the device)
        assert a[key]['depth'].dtype == np.uint16 and b[key]['depth'].dtype == np.float32
        assert np.array_equal(a[key]['depth'].astype(np.float32) / np.float32(256.), b[key]['depth'])
        assert len(a[key]['drops']) > 20 and a[key]['drops'].tobytes() == b[key]['drops'].tobytes()
    # inputs against the loaders themselves, drop tables against pack_frame
    db = gen_n.db
    tables = [f.table for f in db.streaks_simulator.values()]
    for i in range(n):
        if i == 2:
            continue
        bg = imgops.imread_bgr(os.path.join(img_dir, '%06d.png' % i))
        fr = a[bg.tobytes()]
        assert np.array_equal(fr['depth'], imgops.imread_unchanged(os.path.join(dep_dir, '%06d.png' % i)))
        want = hb.pack_frame(tables[i % len(tables)].take(slice(None)), db, 80, 48, i)
        assert fr['drops'].tobytes() == want.tobytes(), i
    # every file from its own frame's scanlines, in both routes
    for out in (out_n, out_g):
        for i in range(n):
            p = os.path.join(out, 'rainy_image', '%06d.png' % i)
            if i == 2:
                assert not os.path.exists(p)
                continue
            bg = imgops.imread_bgr(os.path.join(img_dir, '%06d.png' % i))
            img = np.array(Image.open(p))
            assert img.shape == (48, 80, 4) and (img == np.append(bg[0, 0], 255)).all(), i
            msk = np.array(Image.open(os.path.join(out, 'rain_mask', '%06d.png' % i)))
            assert (msk[..., 1] == bg[0, 0, 1]).all() and (msk[..., 2] == 7).all()
    assert sorted(s['file'][len(out_n):] for s in gen_n.stats) == sorted(s['file'][len(out_g):] for s in gen_g.stats)


def test_native_route_is_not_taken_when_its_conditions_fail(tmp_path, built, monkeypatch):
    """Angular noise, environment-map files, a resized render or a depth map of another size keep the general route."""
    tmp = str(tmp_path)
    src, img_dir, dep_dir = _dataset(tmp, 3)
    monkeypatch.setenv('RAIN_BATCH', '2')
    monkeypatch.setattr(hb, 'RainHip', RecordingContext)
    base = ['--dataset', 'kitti', '-k', src, '-d', src, '-r', os.path.join(tmp, 'particles'), '-sd', os.path.join(tmp, 'rainstreakdb'),
            '-i', '5', '--noverbose']
    for k, extra in enumerate((['--noise_scale', '1', '--noise_std', '3'], ['--save_envmap'])):
        RecordingContext.log = []
        gen = main_mod.main(base + ['--output', os.path.join(tmp, 'o%d' % k)] + extra)
        assert gen.timing[0].get('route') != 'native' and len(RecordingContext.log) == 3
    RecordingContext.log = []
    gen = main_mod.main(base + ['--output', os.path.join(tmp, 'o9')])
    assert gen.timing[0].get('route') == 'native' and len(RecordingContext.log) == 3


def test_native_route_at_render_scale_two(tmp_path, built, monkeypatch):
    """A plug-in that renders at half resolution (what config/cityscapes.py does by default): the batch-native route
    resizes image and depth in the library (rr_io_read_frames_scaled) and hands the GPU the float64 frames, float32
    depth maps and drop tables the general route's per-frame loader makes."""
    tmp = str(tmp_path)
    n = 5
    src, img_dir, dep_dir = _dataset(tmp, n, H=96, W=160)
    kitti = importlib.import_module('rain-rendering_amd.config.kitti')
    plain = kitti.settings

    def half():
        st = dict(plain())
        st["render_scale"] = 2
        return st
    monkeypatch.setattr(kitti, 'settings', half)
    gen_n, log_n, out_n = _run(tmp, src, 'out_native', True, monkeypatch, batch=2)
    gen_g, log_g, out_g = _run(tmp, src, 'out_general', False, monkeypatch, batch=2)
    assert gen_n.timing[0].get('route') == 'native' and gen_g.timing[0].get('route') != 'native'
    assert len(log_n) == len(log_g) == n
    a = {fr['bg'].tobytes(): fr for fr in log_n}
    b = {fr['bg'].tobytes(): fr for fr in log_g}
    assert a.keys() == b.keys() and len(a) == n
    for key in a:
        assert a[key]['bg'].dtype == np.float64 and a[key]['bg'].shape == (48, 80, 3)
        assert a[key]['depth'].dtype == np.float32 and np.array_equal(a[key]['depth'], b[key]['depth'])
        assert len(a[key]['drops']) > 10 and a[key]['drops'].tobytes() == b[key]['drops'].tobytes()
    for i in range(n):
        for out in (out_n, out_g):
            assert np.array(Image.open(os.path.join(out, 'rainy_image', '%06d.png' % i))).shape == (48, 80, 4)


def _rank_worker(rank, world, port, tmp, src, q):
    """One rank of a two-rank driver run (gloo, no GPU: the stand-in context)."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), LOCAL_WORLD_SIZE=str(world),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RAIN_BATCH='2', RAIN_NATIVE_IO='1')
    sys.path.insert(0, os.path.join(h.ROOT, 'scripts'))
    import driver_host_only as dho_
    hb_ = importlib.import_module('rain-rendering_amd.hip_backend')
    hb_.RainHip = dho_.HostOnlyContext
    main_ = importlib.import_module('rain-rendering_amd.main')
    argv = ['--dataset', 'kitti', '-k', src, '-d', src, '-r', os.path.join(tmp, 'particles'), '-sd', os.path.join(tmp, 'rainstreakdb'),
            '-i', '5', '--output', os.path.join(tmp, 'out'), '--noverbose', '--conflict_strategy', 'rename_folder']
    gen = main_.main(argv)
    q.put((rank, sorted(s['file'] for s in gen.stats), gen.timing[0].get('route')))


def test_two_ranks_share_one_run(tmp_path, built):
    """The whole driver under two ranks (gloo; the stand-in context instead of a GPU): rank 0 decides the output folder
    and the work list and loads the streak database, the broadcast hands them over, each rank renders its share on the
    batch-native route, and together they write every frame of the sequence exactly once into ONE folder."""
    import socket
    import torch.multiprocessing as mp
    tmp = str(tmp_path)
    n = 7
    src, img_dir, dep_dir = _dataset(tmp, n)
    os.makedirs(os.path.join(tmp, 'out', 'kitti', 'data_object', 'training', 'rain', '5mm'))    # exists already: 'rename_folder' must pick ONE new name
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, tmp, src, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, f0, route0), (_, f1, route1) = res
    assert route0 == route1 == 'native'
    assert len(f0) + len(f1) == n and not set(f0) & set(f1) and abs(len(f0) - len(f1)) <= 1
    folders = {os.path.dirname(os.path.dirname(f)) for f in f0 + f1}
    assert len(folders) == 1 and folders.pop().endswith('5mm_copy00000')
    for f in f0 + f1:
        assert np.array(Image.open(f)).shape == (48, 80, 4)
        assert os.path.exists(f.replace('rainy_image', 'rain_mask'))


def test_device_particles_mode_sends_the_generator_settings(tmp_path, built, monkeypatch):
    """`--device_particles` (BASELINE configs[4] from the command line): no particle file is read or written; every frame's
    descriptor carries the rr_sim_frame record of its simulated frame (tools/particles.sim_frames: what simulate() would have
    put into the XML file) with the frame's own draw seed, no drop table is made on the host, and the counts the library
    reports end up in the statistics."""
    tmp = str(tmp_path)
    n = 5
    src, img_dir, dep_dir = _dataset(tmp, n)
    import shutil
    shutil.rmtree(os.path.join(tmp, 'particles'))                  # there is no particle file -- and none may appear
    monkeypatch.setenv('RAIN_BATCH', '2')
    monkeypatch.setattr(hb, 'RainHip', RecordingContext)
    RecordingContext.log = []
    tables = []
    monkeypatch.setattr(RecordingContext, 'set_particle_tables', lambda self, dgrid, cdf: tables.append((np.array(dgrid), np.array(cdf))), raising=False)
    gen = main_mod.main(['--dataset', 'kitti', '-k', src, '-d', src, '-r', os.path.join(tmp, 'particles'), '-sd', os.path.join(tmp, 'rainstreakdb'),
                         '-i', '5', '--output', os.path.join(tmp, 'out'), '--noverbose', '--device_particles'])
    assert not os.path.exists(os.path.join(tmp, 'particles'))
    assert gen.timing[0].get('route') == 'native' and len(RecordingContext.log) == n and len(tables) == 1
    particles = importlib.import_module('rain-rendering_amd.tools.particles')
    db = importlib.import_module('rain-rendering_amd.common.db')
    opts = db.sim('kitti', 'data_object/training', os.path.join(tmp, 'particles', 'kitti'))['options']
    n_sim = particles.n_sim_frames(opts)
    want, dgrid, cdf = particles.sim_frames(opts, 5, n_sim, render_scale=1, seed=0)
    assert np.array_equal(tables[0][0], dgrid) and np.array_equal(tables[0][1], cdf)
    seen = set()
    for fr in RecordingContext.log:
        f = int(fr['sim']['draw_seed'])
        seen.add(f)
        ref = want[f % n_sim].copy()
        ref['draw_seed'] = f
        assert fr['sim'].tobytes() == ref.tobytes(), f
    assert seen == set(range(n))
    assert sorted(s['drops'] for s in gen.stats) == [100 + f for f in range(n)]
    for i in range(n):
        assert os.path.exists(os.path.join(tmp, 'out', 'kitti', 'data_object', 'training', 'rain', '5mm', 'rainy_image', '%06d.png' % i))
