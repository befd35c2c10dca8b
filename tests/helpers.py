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

scenes = importlib.import_module("rain-rendering_amd.scenes")
KITTI, CITYSCAPES, NUSCENES = scenes.KITTI, scenes.CITYSCAPES, scenes.NUSCENES


class Scene(scenes.Scene):
    """The product-side synthetic sequence (rain-rendering_amd/scenes.py) plus the ORACLE loaders of the
    same files, so a test can feed identical inputs to both."""

    def oracle_streaks(self, i):
        sim = orc.load_streaks_from_xml(self.xml, self.render_scale, [self.W, self.H])
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
    hdrs = [os.path.join(ROOT, 'rain-rendering_amd', 'csrc', h) for h in ('rr_device.h', 'rr_prepass.h', 'rr_deflate.h', 'rr_pngrows.h')]
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


def emu_render(scene, bg, rainy_bg, env_xyY, drops, opacity=1.0, strategy=0, depth=None):
    emu = hostemu()
    texels, hs, ws, offs = hb.pack_streak_db(scene.db.streaks_light)
    H, W = bg.shape[:2]
    n = len(drops)
    out = dict(image_u8=np.zeros((H, W, 3), np.uint8), rainy_bg=np.zeros((H, W, 3)), mask=np.zeros((H, W)),
               mask_i32=np.zeros((H, W), np.int32), status=np.zeros(max(n, 1), np.int32), K=np.zeros((max(n, 1), 3)))
    drops = np.ascontiguousarray(drops)
    if depth is not None:
        depth = np.ascontiguousarray(depth, np.float32 if np.asarray(depth).dtype == np.float32 else np.float64)
        assert depth.shape == (H, W)
    emu.emu_render_frame_depth(H, W, scene.He, scene.We, _p(bg), _p(rainy_bg), _p(env_xyY), _p(scene.omega), _p(drops), n,
                               ctypes.byref(scene.cam), ctypes.c_double(opacity), _p(texels), _p(hs), _p(ws), _p(offs),
                               _p(out['image_u8']), _p(out['rainy_bg']), _p(out['mask']), _p(out['mask_i32']), _p(out['status']),
                               _p(out['K']), int(strategy), _p(depth) if depth is not None else None,
                               1 if depth is not None and depth.dtype == np.float64 else 0)
    out['status'] = out['status'][:n]
    return out


def oracle_render(scene, i, bg, rainy_bg, env_xyY, faithful=True, noise_std=0.0, noise_scale=0.0, max_drops=None, strategy=None,
                  first_drop=0, scene_depth=None):
    textures, ratio = scene.oracle_db()
    streaks = scene.oracle_streaks(i)
    return orc.render_frame(bg, rainy_bg, env_xyY, scene.omega, streaks, textures, ratio, scene.ocam, frame_seed=i,
                            noise_std=noise_std, noise_scale=noise_scale, faithful=faithful, max_drops=max_drops,
                            rendering_strategy=strategy, first_drop=first_drop, scene_depth=scene_depth)


def prepass_scene(H, W, seed, dtype=np.float32):
    """Seeded inputs of the pre-pass tests (image / 255 and a depth map in metres); tests/golden/make_golden_prepass.py
    feeds the same arrays to the reference."""
    bg = synthetic.make_frame(seed, H, W)
    rng = np.random.RandomState(seed)
    depth = (np.linspace(80, 2, H)[:, None] * np.ones((1, W)) + rng.uniform(0, 3, (H, W))).astype(dtype)
    return bg, depth
