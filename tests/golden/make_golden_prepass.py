"""Generates tests/golden/prepass_vectors.npz by IMPORTING THE REFERENCE (read-only,
/root/reference) in the build container and recording inputs + outputs of its own pre-pass
code on seeded inputs (the reference never travels to the GPU box; only the vectors do).

    python tests/golden/make_golden_prepass.py

Run through the reference's own code (=> these pin oracle/prepass.py):
  FogRain.fog_rain_layer        common/add_attenuation.py:26-95  (extinction map, irradiance mean,
                                l_in, clips, final composition -- every numpy operation)
  EnvironmentMapGenerator.generate_map   common/bad_weather.py:742-819  (cylinder projection incl.
                                np.unique's first-pixel rule, the per-pixel fill_matrices loops, the
                                mirrored sides, the masked replacement)
cv2 is not installed; on top of make_golden.py's shims this script adds
  * cv2.flip          -> np.flip                      (definitionally identical)
  * cv2.GaussianBlur  -> oracle.prepass.gaussian_blur (float) / round-half-even of it (uint8)
                         NOT the real library: the blur arithmetic itself stays UNPINNED (OpenCV's
                         uint8 path is fixed-point), every operation around it is pinned.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # noqa: E402  (shims + paths)


def main():
    mg.install_shims()
    from oracle import prepass as op
    cv2 = sys.modules['cv2']

    def flip(a, code):
        return np.flip(a, 0 if code == 0 else 1)

    def GaussianBlur(img, ksize, sigma):
        if img.dtype == np.uint8:
            return np.clip(np.rint(op.gaussian_blur(img.astype(np.float64), ksize[0], sigma)), 0, 255).astype(np.uint8)
        return op.gaussian_blur(img, ksize[0], sigma)

    cv2.flip = flip
    cv2.GaussianBlur = GaussianBlur
    os.chdir(mg.REF)
    from common import add_attenuation as radd, bad_weather as rbw
    import helpers as h

    out = {}
    cases = [(48, 80, 25, np.float32, 3), (41, 67, 100, np.float64, 4)]      # even and odd sizes, both depth dtypes
    for k, (H, W, rain, dtype, seed) in enumerate(cases):
        bg = h.synthetic.make_frame(seed, H, W)
        rng = np.random.RandomState(seed)
        depth = (np.linspace(80, 2, H)[:, None] * np.ones((1, W)) + rng.uniform(0, 3, (H, W))).astype(dtype)
        fog = radd.FogRain(rain_intensity=rain, focal=0.006, f_number=6.0, angle=90, exposure=2, camera_gain=20)
        rainy = fog.fog_rain_layer(bg.copy(), depth.copy())
        env = rbw.EnvironmentMapGenerator(0.006, W, H).generate_map(rainy.copy())
        out['case%d_meta' % k] = np.array([H, W, rain, seed], np.int64)
        out['case%d_bg' % k] = bg
        out['case%d_depth' % k] = depth
        out['case%d_rainy' % k] = rainy
        out['case%d_env' % k] = env
    # ---- KITTI size (1242x375), fog -> environment map end to end through the reference's classes.  The inputs are
    # regenerated from the seed by the tests; the outputs are recorded as SHA-256 digests of the full float64 arrays
    # (the oracle must reproduce them bit for bit) plus every 25th row for diagnosis.
    import hashlib
    H, W, rain, seed = 375, 1242, 50, 3
    bg = h.synthetic.make_frame(seed, H, W)
    rng = np.random.RandomState(seed)
    depth = (np.linspace(80, 2, H)[:, None] * np.ones((1, W)) + rng.uniform(0, 3, (H, W))).astype(np.float32)
    fog = radd.FogRain(rain_intensity=rain, focal=0.006, f_number=6.0, angle=90, exposure=2, camera_gain=20)
    rainy = fog.fog_rain_layer(bg.copy(), depth.copy())
    env = rbw.EnvironmentMapGenerator(0.006, W, H).generate_map(rainy.copy())
    out['kitti_meta'] = np.array([H, W, rain, seed], np.int64)
    out['kitti_env_shape'] = np.array(env.shape, np.int64)
    out['kitti_digests'] = np.array([hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() for a in (bg, depth, rainy, env)])
    out['kitti_rainy_rows'] = rainy[::25]
    out['kitti_env_rows'] = env[::25]
    np.savez_compressed(os.path.join(HERE, 'prepass_vectors.npz'), **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == '__main__':
    main()
