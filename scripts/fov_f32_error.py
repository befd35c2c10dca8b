"""Error budget of float32 environment-map sums (RR_OPT_FOV_F32, k_fov_sums32) on a KITTI-shaped scene, without a GPU:
the prefix rows of (x*w, y*w, Y*w, w) and the per-row span differences are formed in float32 the way the kernel does
(inclusive prefix per row, P[xr + 1] - P[xl], float running sums over the rows), against the same sums in float64; the
spans are a drop-like family of column intervals (random centre, 40-90 % of the map's width, every row).  Prints the
relative error of the four sums and of the colour ratios x = Sx/Sw, y = Sy/Sw, Y = SY/Sw the drop's colour is made of.

    python scripts/fov_f32_error.py [--drops 2000]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--drops', type=int, default=2000)
    ap.add_argument('--He', type=int, default=375)
    ap.add_argument('--We', type=int, default=1909)
    args = ap.parse_args()
    import importlib
    solid_angle = importlib.import_module('rain-rendering_amd.common.solid_angle')
    rng = np.random.RandomState(0)
    He, We = args.He, args.We
    env = rng.rand(He, We, 3)
    for _ in range(2):                                    # image-like: low-pass
        env = (env + np.roll(env, 1, 0) + np.roll(env, -1, 0) + np.roll(env, 1, 1) + np.roll(env, -1, 1)) / 5
    env[..., 2] *= 3.0                                    # Y
    omega = solid_angle.get_solid_angles(np.empty((He, We, 0)))
    comp64 = np.stack([env[..., 0] * omega, env[..., 1] * omega, env[..., 2] * omega, omega], axis=-1)     # [He][We][4]
    comp32 = comp64.astype(np.float32)
    P64 = np.concatenate([np.zeros((He, 1, 4)), np.cumsum(comp64, axis=1)], axis=1)
    P32 = np.concatenate([np.zeros((He, 1, 4), np.float32), np.cumsum(comp32, axis=1, dtype=np.float32)], axis=1)
    worst = np.zeros(4)
    worst_ratio = np.zeros(3)
    for _ in range(args.drops):
        width = int(We * rng.uniform(0.4, 0.9))
        xl = rng.randint(0, We - width, He)               # a different interval per row, like a polygon's spans
        xr = xl + width + rng.randint(-20, 20, He)
        xr = np.clip(xr, xl, We - 1)
        rows = np.arange(He)
        s64 = (P64[rows, xr + 1] - P64[rows, xl]).sum(axis=0)
        d32 = P32[rows, xr + 1] - P32[rows, xl]
        s32 = np.zeros(4, np.float32)
        for y in range(He):                               # float running sums, row after row
            s32 = s32 + d32[y]
        worst = np.maximum(worst, np.abs(s32.astype(np.float64) - s64) / np.abs(s64))
        r64, r32 = s64[:3] / s64[3], s32[:3].astype(np.float64) / float(s32[3])
        worst_ratio = np.maximum(worst_ratio, np.abs(r32 - r64) / np.abs(r64))
    print("float32 vs float64 over %d span families on a %dx%d map:" % (args.drops, We, He))
    print("  largest relative error of the sums (x*w, y*w, Y*w, w): %s" % np.array2string(worst, precision=2))
    print("  largest relative error of the colour ratios (x, y, Y):  %s" % np.array2string(worst_ratio, precision=2))
    print("  one LSB of an 8-bit channel is a relative 3.9e-03 of full scale")


if __name__ == '__main__':
    main()
