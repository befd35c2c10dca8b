"""Shared fixtures for the test tiers: synthetic scene construction in the reference's
on-disk formats, loading through BOTH the product loaders and the oracle loaders, and the
ctypes view of the host-emulation library (tests/hostemu)."""
import ctypes
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pkg = importlib.import_module("rain-rendering_amd")
hb = importlib.import_module("rain-rendering_amd.hip_backend")
synthetic = importlib.import_module("rain-rendering_amd.synthetic")
bw = importlib.import_module("rain-rendering_amd.common.bad_weather")
my_utils = importlib.import_module("rain-rendering_amd.common.my_utils")
solid_angle = importlib.import_module("rain-rendering_amd.common.solid_angle")

from oracle import render as orc  # noqa: E402

KITTI = dict(focal_mm=6.0, f_number=6.0, exposure_ms=2.0, pix_um=4.65)


class Scene:
    """One synthetic sequence: streak DB + particles on disk, frames/envmaps in memory."""

    def __init__(self, tmpdir, H, W, n_drops, n_frames=1, cam=KITTI, seed0=3000, far_fraction=0.02, frames=None,
                 tex_heights=None, tex_width=None):
        self.H, self.W = H, W
        self.cam_settings = cam
        self.tex_dir, self.norm = synthetic.write_streak_db(os.path.join(str(tmpdir), 'rainstreakdb'),
                                                            tex_heights=tex_heights, tex_width=tex_width)
        if frames is None:
            frames = synthetic.simulate_particles(n_frames, n_drops, W, H, cam['focal_mm'], cam['pix_um'], cam['exposure_ms'],
                                                  seed0=seed0, far_fraction=far_fraction)
        self.xml = synthetic.write_particles_xml(os.path.join(str(tmpdir), 'particles', 'rain', 'sim_camera0.xml'), frames)
        self.He = H
        self.We = synthetic.envmap_width(cam['focal_mm'], W)
        # product loaders
        self.db = bw.DBManager(streaks_path=self.tex_dir, streaks_path_xml=self.xml, norm_coeff_path=self.norm)
        self.db.load_streak_database()
        self.db.load_streaks_from_xml('kitti', {"render_scale": 1}, [W, H], use_pickle=False, verbose=False)
        self.omega = solid_angle.get_solid_angles(np.zeros((self.He, self.We)))
        self.cam = hb.make_camera(cam['focal_mm'] / 1000., cam['f_number'], cam['exposure_ms'])
        self.ocam = dict(focal_m=cam['focal_mm'] / 1000., f_number=cam['f_number'], exposure_ms=cam['exposure_ms'])

    def frame_inputs(self, i):
        bg = synthetic.make_frame(i, self.H, self.W)
        env_bgr = synthetic.make_envmap(i, self.He, self.We)
        env_xyY = my_utils.convert_rgb_to_xyY(env_bgr[..., ::-1])
        env_xyY[np.isnan(env_xyY)] = 0
        return bg, np.ascontiguousarray(env_xyY)

    def product_drops(self, i, noise_std=0.0, noise_scale=0.0):
        """What Generator.run does before the GPU call: seed, filter, pack."""
        frames = list(self.db.streaks_simulator.values())
        fr = frames[i % len(frames)]
        np.random.seed(i)
        idx = hb.filter_streaks(fr.table, self.W, self.H)
        return hb.pack_drops(fr.table, idx, self.db, noise_std, noise_scale)

    def oracle_streaks(self, i):
        sim = orc.load_streaks_from_xml(self.xml, 1, [self.W, self.H])
        frames = list(sim.values())
        fr = frames[i % len(frames)]
        return list(orc.streak_filter(fr.streaks, self.W, self.H).values())

    def oracle_db(self):
        return orc.load_streak_database(self.tex_dir, self.norm)


# ---------------------------------------------------------------------------
# host emulation of the kernel arithmetic (tests/hostemu)
# ---------------------------------------------------------------------------
_emu = None


def build_hostemu():
    src = os.path.join(ROOT, 'tests', 'hostemu', 'hostemu.cpp')
    out = os.path.join(ROOT, 'tests', 'hostemu', 'libhostemu.so')
    hdrs = [os.path.join(ROOT, 'rain-rendering_amd', 'csrc', h) for h in ('rr_device.h', 'rr_prepass.h')]
    if (not os.path.exists(out)) or os.path.getmtime(out) < max(os.path.getmtime(f) for f in [src] + hdrs):
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
                               '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'rain-rendering_amd', 'csrc'),
                               src, '-o', out])
    return out


def hostemu():
    global _emu
    if _emu is None:
        _emu = ctypes.CDLL(build_hostemu())
    return _emu


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def emu_render(scene, bg, rainy_bg, env_xyY, drops, opacity=1.0, strategy=0):
    emu = hostemu()
    texels, hs, ws, offs = hb.pack_streak_db(scene.db.streaks_light)
    H, W = bg.shape[:2]
    n = len(drops)
    out = dict(image_u8=np.zeros((H, W, 3), np.uint8), rainy_bg=np.zeros((H, W, 3)), mask=np.zeros((H, W)),
               mask_i32=np.zeros((H, W), np.int32), status=np.zeros(max(n, 1), np.int32), K=np.zeros((max(n, 1), 3)))
    drops = np.ascontiguousarray(drops)
    emu.emu_render_frame(H, W, scene.He, scene.We, _p(bg), _p(rainy_bg), _p(env_xyY), _p(scene.omega), _p(drops), n,
                         ctypes.byref(scene.cam), ctypes.c_double(opacity), _p(texels), _p(hs), _p(ws), _p(offs),
                         _p(out['image_u8']), _p(out['rainy_bg']), _p(out['mask']), _p(out['mask_i32']), _p(out['status']),
                         _p(out['K']), int(strategy))
    out['status'] = out['status'][:n]
    return out


def oracle_render(scene, i, bg, rainy_bg, env_xyY, faithful=True, noise_std=0.0, noise_scale=0.0, max_drops=None, strategy=None):
    textures, ratio = scene.oracle_db()
    streaks = scene.oracle_streaks(i)
    return orc.render_frame(bg, rainy_bg, env_xyY, scene.omega, streaks, textures, ratio, scene.ocam, frame_seed=i,
                            noise_std=noise_std, noise_scale=noise_scale, faithful=faithful, max_drops=max_drops,
                            rendering_strategy=strategy)
