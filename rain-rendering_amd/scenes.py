"""Synthetic sequences in the reference's on-disk formats, loaded through the PRODUCT loaders.

Used by bench.py, __graft_entry__.smoke(), the scripts and (sub-classed with the oracle-side
loaders) by tests/helpers.py.  Nothing here imports oracle/.

Camera presets follow the reference's dataset plug-ins:
  KITTI       config/kitti.py:49-60       (6 mm, f/6, 2 ms, 4.65 um)
  CITYSCAPES  config/cityscapes.py:27-42  (6 mm, f/6, 5 ms, 2.2 um; render_scale = depth_scale = 2)
  NUSCENES    config/nuscenes.py:66-72    (5.5 mm, f/1.8, 5 ms; pixel size from the defaults, db.py:16)
"""
import importlib
import os

import numpy as np

_pkg = __name__.rsplit('.', 1)[0]
hb = importlib.import_module(_pkg + '.hip_backend')
synthetic = importlib.import_module(_pkg + '.synthetic')
bw = importlib.import_module(_pkg + '.common.bad_weather')
my_utils = importlib.import_module(_pkg + '.common.my_utils')
solid_angle = importlib.import_module(_pkg + '.common.solid_angle')

KITTI = dict(focal_mm=6.0, f_number=6.0, exposure_ms=2.0, pix_um=4.65)
CITYSCAPES = dict(focal_mm=6.0, f_number=6.0, exposure_ms=5.0, pix_um=2.2)
NUSCENES = dict(focal_mm=5.5, f_number=1.8, exposure_ms=5.0, pix_um=4.65)


class Scene:
    """One synthetic sequence: streak DB + particles on disk, frames/envmaps in memory.

    H, W are the RENDERED frame size.  With render_scale = s the particles are simulated on the
    s*W x s*H sensor and the loader divides image coordinates and diameters by s
    (reference bad_weather.py:208-211), exactly as the Cityscapes plug-in does (s = 2)."""

    def __init__(self, tmpdir, H, W, n_drops, n_frames=1, cam=KITTI, seed0=3000, far_fraction=0.02, frames=None,
                 tex_heights=None, tex_width=None, render_scale=1, dataset='kitti'):
        self.H, self.W = H, W
        self.cam_settings = cam
        self.render_scale = render_scale
        self.tex_dir, self.norm = synthetic.write_streak_db(os.path.join(str(tmpdir), 'rainstreakdb'),
                                                            tex_heights=tex_heights, tex_width=tex_width)
        xml = os.path.join(str(tmpdir), 'particles', 'rain', 'sim_camera0.xml')
        # a directory that already holds this very simulation (same parameters: the stamp) is reused -- bench.py's
        # counter passes run in child processes and would otherwise simulate and format the same 2 M streaks again
        stamp = repr((n_frames, n_drops, W, H, render_scale, seed0, far_fraction, sorted(cam.items()))) if frames is None else None
        stamp_path = xml + '.stamp'
        if stamp is not None and os.path.exists(xml) and os.path.exists(stamp_path) and open(stamp_path).read() == stamp:
            self.xml = xml
        else:
            if frames is None:            # (frame by frame on a pool of processes: the same bytes as simulate + write)
                self.xml = synthetic.simulate_to_xml(xml, n_frames, n_drops, W * render_scale, H * render_scale, cam['focal_mm'],
                                                     cam['pix_um'], cam['exposure_ms'], seed0=seed0, far_fraction=far_fraction)
            else:
                self.xml = synthetic.write_particles_xml(xml, frames)
            if stamp is not None:
                with open(stamp_path, 'w') as fh:
                    fh.write(stamp)
        self.He = H
        self.We = synthetic.envmap_width(cam['focal_mm'], W)
        # product loaders
        self.db = bw.DBManager(streaks_path=self.tex_dir, streaks_path_xml=self.xml, norm_coeff_path=self.norm)
        self.db.load_streak_database()
        self.db.load_streaks_from_xml(dataset, {"render_scale": render_scale}, [W, H], use_pickle=False, verbose=False)
        self.omega = solid_angle.get_solid_angles(np.zeros((self.He, self.We)))
        self.cam = hb.make_camera(cam['focal_mm'] / 1000., cam['f_number'], cam['exposure_ms'])
        self.ocam = dict(focal_m=cam['focal_mm'] / 1000., f_number=cam['f_number'], exposure_ms=cam['exposure_ms'])

    def frame_inputs(self, i):
        bg = synthetic.make_frame(i, self.H, self.W)
        env_bgr = synthetic.make_envmap(i, self.He, self.We)
        env_xyY = my_utils.convert_rgb_to_xyY(env_bgr[..., ::-1])
        env_xyY[np.isnan(env_xyY)] = 0
        return bg, np.ascontiguousarray(env_xyY)

    def product_drops(self, i, noise_std=0.0, noise_scale=0.0, seed=None):
        """What Generator.run does before the GPU call: seed, filter, pack."""
        frames = list(self.db.streaks_simulator.values())
        fr = frames[i % len(frames)]
        np.random.seed(i if seed is None else seed)
        idx = hb.filter_streaks(fr.table, self.W, self.H)
        return hb.pack_drops(fr.table, idx, self.db, noise_std, noise_scale)
