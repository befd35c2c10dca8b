"""GPU tier: drop tables born on the device (SURVEY 8f "next" #4; BASELINE.json configs[4]: nuScenes 1600x900,
{1, 5, 25, 100, 200} mm/hr, "in-kernel particle simulation (no XML)").

  1. k_particles + k_particle_draws == their host statement (tools/particles.py: make_particles -> the loader's derived
     fields -> frame filter + the legacy-stream draws), every field of every rr_drop record bit for bit;
  2. nuScenes frames rendered from device-generated records -- no XML file, no host drop table, nothing uploaded but the
     generator's settings -- against the g++ build of the kernel arithmetic at full size and the numpy oracle on windows,
     both fed the very records the device made (downloaded for the purpose): mask bit-exact, image <= 1 LSB.
There is no reference oracle for the simulation itself (the reference's simulator is a closed binary); what is checked
is that the device does what the build's own definition says, and that the frames rendered from it obey the parity bar."""
import importlib

import numpy as np
import pytest

import helpers as h
from oracle import render as orc

pytestmark = pytest.mark.gpu

particles = importlib.import_module('rain-rendering_amd.tools.particles')
db = importlib.import_module('rain-rendering_amd.common.db')


def _options(dataset, **kw):
    o = dict(db.settings(dataset))
    o.pop('sequences', None)
    o.update(kw)
    return o


def _rh(sc):
    rh = h.hb.RainHip(0)
    rh.set_streak_db(sc.db.streaks_light)
    rh.set_camera(sc.cam)
    return rh


@pytest.mark.parametrize("dataset,rs,rate,count", [('nuscenes', 1, 1, None), ('nuscenes', 1, 5, None), ('nuscenes', 1, 25, None),
                                                   ('nuscenes', 1, 100, None), ('nuscenes', 1, 200, None), ('nuscenes', 1, 200, 16384),
                                                   ('kitti', 1, 100, None), ('cityscapes', 2, 50, None)])
def test_device_records_equal_host_statement(tmp_path, built, dataset, rs, rate, count):
    sc = h.Scene(tmp_path, 64, 96, 10)                       # (only its streak database is used: the texture ratios)
    opt = _options(dataset, sim_steps={"cam_motion": np.array([30.0, 50.0, 0.0])})
    nf = 3
    sims, dgrid, cdf = particles.sim_frames(opt, rate, nf, render_scale=rs, seed=1234 + 2 ** 40, draw_seeds=[0, 17, 2 ** 31 + 5], count=count)
    want = particles.expected_records(sims, dgrid, cdf, sc.db)
    W, H = opt["cam_CCD_WH"][0] // rs, opt["cam_CCD_WH"][1] // rs
    rh = _rh(sc)
    try:
        rh.set_particle_tables(dgrid, cdf)
        got, cnt = rh.generate_drops(sims, H, W)
        # a capacity below the drop count: the count still tells, the records that fit are the first ones
        small, cnt_small = rh.generate_drops(sims, H, W, cap=max(len(want[0]) // 2, 1))
    finally:
        rh.close()
    for k in range(nf):
        assert int(cnt[k]) == len(want[k]) == len(got[k]), 'frame %d: %d drops on the device, %d on the host' % (k, cnt[k], len(want[k]))
        for name in h.hb.DROP_DTYPE.names:
            assert got[k][name].tobytes() == want[k][name].tobytes(), '%s, frame %d' % (name, k)
    assert np.array_equal(cnt_small, cnt)
    assert small[0].tobytes() == want[0][:len(small[0])].tobytes()
    if rate >= 25:
        assert set(np.concatenate([w['type'] for w in want])) == {0, 1, 2}


# fall rate -> particles per frame: the physical (Poisson) count, and for the heaviest rain also SURVEY 8d's fixed 16384
@pytest.mark.parametrize("rate,count,windows", [(1, None, [(0, 10 ** 6)]), (5, None, [(0, 300)]), (25, None, [(200, 500)]),
                                                (100, None, [(0, 200), (3000, 3200)]), (200, None, [(5000, 5300)]),
                                                (200, 16384, [(9000, 9300)])])
def test_nuscenes_rendered_from_device_generated_records(tmp_path, built, rate, count, windows):
    H, W = 900, 1600
    sc = h.Scene(tmp_path, H, W, 10, cam=h.NUSCENES)         # streak database, camera, environment-map geometry; its XML is not used
    opt = _options('nuscenes')
    sims, dgrid, cdf = particles.sim_frames(opt, rate, 2, seed=77, draw_seeds=[5, 6], count=count)
    bg, env = sc.frame_inputs(0)
    textures, _ = sc.oracle_db()
    rh = _rh(sc)
    try:
        rh.set_particle_tables(dgrid, cdf)
        # THE path under test: settings in, image out; the drop table never exists on the host
        out = rh.render_frames([dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, sim=sims[1])])[0]
        # for the checkers: the same table, downloaded
        rec = rh.generate_drops(sims[1:2], H, W)[0][0]
        assert out['n_drops'] == len(rec) and len(rec) > (3 if rate == 1 else 50)
        emu = h.emu_render(sc, bg, bg, env, rec)
        assert np.array_equal(out['status'], emu['status'])
        assert np.array_equal(out['mask'], emu['mask']) and np.array_equal(out['mask_i32'], emu['mask_i32'])
        assert np.abs(out['image_u8'].astype(int) - emu['image_u8'].astype(int)).max() <= 1
        assert (out['status'] == 0).sum() > 0.9 * len(rec) and out['mask'].max() > 0
        for a, b in windows:
            b = min(b, len(rec))
            win = rh.render_frames([dict(bg=bg, rainy_bg=bg, env_xyY=env, omega=sc.omega, drops=rec[a:b])])[0]
            ref = orc.render_drop_records(bg, bg, env, sc.omega, rec[a:b], textures, sc.ocam, faithful=True)
            assert np.array_equal(win['status'], ref['status'])
            assert np.array_equal(win['mask'], ref['mask']) and np.array_equal(win['mask_i32'], ref['mask_i32'])
            assert np.abs(win['image_u8'].astype(int) - ref['image_u8'].astype(int)).max() <= 1
    finally:
        rh.close()


def test_generated_tables_through_the_async_pipeline(tmp_path, built):
    """rr_pipeline_submit with rr_frame_in.sim: pre-pass + generator + hot path, several frames per call, counts returned."""
    fogmod = importlib.import_module('rain-rendering_amd.common.add_attenuation')
    envmod = importlib.import_module('rain-rendering_amd.common.envmap')
    imgops = importlib.import_module('rain-rendering_amd.common.imgops')
    H, W = 225, 400
    sc = h.Scene(tmp_path, H, W, 10, cam=h.NUSCENES)
    opt = _options('nuscenes', cam_CCD_WH=[W, H])
    nf = 4
    sims, dgrid, cdf = particles.sim_frames(opt, 100, nf, seed=3, count=900)
    want = particles.expected_records(sims, dgrid, cdf, sc.db)
    cs = sc.cam_settings
    consts = fogmod.FogRain(rain_intensity=100, focal=cs['focal_mm'] / 1000., f_number=cs['f_number'], angle=90, exposure=cs['exposure_ms'],
                            camera_gain=1.0).constants()
    rh = _rh(sc)
    try:
        rh.set_particle_tables(dgrid, cdf)
        rh.set_prepass_kernels(imgops.gaussian_kernel(25, 25), imgops.gaussian_kernel(15, 0))
        we = rh.set_envmap_geometry(H, W, *envmod.EnvironmentMapGenerator(cs['focal_mm'] / 1000., W, H).device_tables(H, W))
        omega = h.solid_angle.get_solid_angles(np.empty((H, we, 0)))
        bgs = [np.ascontiguousarray((h.synthetic.make_frame(i, H, W) * 255).astype(np.uint8)) for i in range(nf)]
        depth = (np.linspace(80, 2, H, dtype=np.float32)[:, None] * np.ones((1, W), np.float32))
        gen = [dict(bg_u8=bgs[i], depth=depth, fog=consts, omega=omega, sim=sims[i], drops_cap=900) for i in range(nf)]
        ref = [dict(bg_u8=bgs[i], depth=depth, fog=consts, omega=omega, drops=want[i]) for i in range(nf)]
        outs_g = [dict(image_u8=np.zeros((H, W, 3), np.uint8), mask=np.zeros((H, W)), status=np.zeros(900, np.int32), n_drops=np.zeros(1, np.int32))
                  for _ in range(nf)]
        outs_r = [dict(image_u8=np.zeros((H, W, 3), np.uint8), mask=np.zeros((H, W)), status=np.zeros(len(want[i]), np.int32)) for i in range(nf)]
        for slot, (frs, outs) in enumerate(((gen, outs_g), (ref, outs_r))):
            rh.pipeline_submit(slot, frs, outs)
        for slot in range(2):
            while not rh.pipeline_wait(slot):
                rh.pipeline_submit(slot, *((gen, outs_g), (ref, outs_r))[slot])
    finally:
        rh.close()
    for i in range(nf):
        n = int(outs_g[i]['n_drops'][0])
        assert n == len(want[i]) > 100
        assert np.array_equal(outs_g[i]['status'][:n], outs_r[i]['status'])
        assert np.array_equal(outs_g[i]['mask'], outs_r[i]['mask']) and np.array_equal(outs_g[i]['image_u8'], outs_r[i]['image_u8'])
        assert outs_g[i]['mask'].max() > 0
