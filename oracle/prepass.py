"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the two per-frame pre-passes that produce the
hot path's inputs (SURVEY 8f "next" #1 and #2), restated from the reference in numpy:

  fog_rain_layer     reference common/add_attenuation.py:26-95  (FogRain)
  generate_env_map   reference common/bad_weather.py:707-853     (EnvironmentMapGenerator)
  env_to_xyY         reference common/generator.py:407-408

PINNED by tests/golden/prepass_vectors.npz (tests/golden/make_golden_prepass.py runs the
reference's own FogRain.fog_rain_layer and EnvironmentMapGenerator.generate_map in the build
container; tests/test_oracle_pin.py reproduces them bit for bit) -- EXCEPT the arithmetic of
cv2.GaussianBlur itself, which stays PARITY UNPINNED: cv2 is not installed, so the generator
substitutes gaussian_blur() below for it.  The kernels follow cv::getGaussianKernel, borders
are BORDER_REFLECT_101, the uint8 blur rounds half to even (OpenCV's uint8 path is
fixed-point and may differ by 1 LSB in the filled-in parts of the environment map).  Data
types follow the reference's numpy arithmetic (float32 depth -> float32 extinction map)."""
import math

import numpy as np
from scipy.ndimage import correlate1d

from .render import convert_rgb_to_xyY


def gaussian_kernel(ksize, sigma):
    """cv::getGaussianKernel(ksize, sigma)."""
    if sigma <= 0:
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = np.arange(ksize) - (ksize - 1) / 2.0
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return k / k.sum()


def gaussian_blur(img, ksize, sigma):
    k = gaussian_kernel(ksize, sigma)
    out = correlate1d(img, k, axis=1, mode='mirror')
    return correlate1d(out, k, axis=0, mode='mirror')


def fog_constants(rain_intensity, f_number, exposure_ms, camera_gain, angle=90):
    beta_ext = 0.312 * rain_intensity ** 0.67                                            # add_attenuation.py:40-43
    g = 0.97
    cos_term = math.cos(math.radians(angle))
    beta_hg = (1 - (g ** 2)) / (4 * np.pi * ((1 + g ** 2 - 2 * g * cos_term) ** 1.5))  # :60-64
    irr_scale = (4 * (f_number ** 2)) / ((exposure_ms * 1e-3) * camera_gain * np.pi)    # :51-54
    return beta_ext, beta_hg, irr_scale


def fog_rain_layer(image, depth, rain_intensity, f_number, exposure_ms, camera_gain, angle=90):
    """add_attenuation.py:88-95 -> calc_l :75-86."""
    beta_ext, beta_hg, _ = fog_constants(rain_intensity, f_number, exposure_ms, camera_gain, angle)
    f_ext = np.exp((-beta_ext) * (depth / 1000))                                         # :48
    f_ext = np.tile(np.expand_dims(f_ext, axis=-1), (1, 1, 3))
    irradiance = (4 * (f_number ** 2) * image) / ((exposure_ms * 1e-3) * camera_gain * np.pi)
    irradiance_mean = np.mean(irradiance.reshape(-1, 3), axis=0)
    l_in = np.clip(beta_hg * irradiance_mean * (1 - f_ext), 0, 1)                        # :66-73
    f_ext = gaussian_blur(f_ext, 25, 25)                                                 # :79-80
    l_in = gaussian_blur(l_in, 25, 25)
    return np.clip(np.clip(image * f_ext + l_in, 0, 1), 0, 1)                            # :85-86,93


def env_geometry(focal_m, H, W):
    """Projection tables of EnvironmentMapGenerator for an HxW frame (bad_weather.py:712,716-761)."""
    focal = int(((focal_m * 1000) / 12.7) * W)
    center = np.array([int(W // 2), int(H // 2)])
    max_x = round(focal * np.arctan(center[0] / focal) + center[0])
    min_x = round(focal * np.arctan(-center[0] / focal) + center[0])
    cw = int(max_x - min_x) + 1
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing='ij')
    d_row, d_col = yy - center[1], xx - center[0]
    rows = np.round(focal * (d_row / np.sqrt(d_col ** 2 + focal ** 2)) + center[1])
    cols = np.round(focal * np.arctan(d_col / focal) + center[0]) - min_x
    key = rows.astype(np.int32).astype(np.int64).ravel() * cw + cols.astype(np.int32).astype(np.int64).ravel()
    uniq, first = np.unique(key, return_index=True)          # == np.unique(xy, axis=0, return_index=True) (:762)
    return cw, uniq, first


def generate_env_map(background, focal_m):
    """bad_weather.py:742-819; returns the BGR float map in [0,1]."""
    bg8 = (background * 255).astype(np.uint8)
    H, W = bg8.shape[:2]
    cw, uniq, first = env_geometry(focal_m, H, W)
    cyl = np.zeros((H, cw, 3), np.uint8)
    cyl.reshape(-1, 3)[uniq] = bg8.reshape(-1, 3)[first]
    mask = np.zeros((H, cw), np.uint8)
    mask.reshape(-1)[uniq] = 255
    half = H // 2
    # fill_matrices :821-853 + :776-789: unfilled pixels take the column's first filled pixel
    # seen from the bottom (bottom half) / from the top (top half)
    fl, mfl = cyl[::-1], mask[::-1]
    tmp = fl[:half].copy()
    r, c = np.nonzero(mfl[:half] == 0)
    src = np.argmax(mask[half:][::-1] > 0, axis=0)
    tmp[r, c] = fl[src[c], c]
    if half:
        cyl[-half:] = tmp[::-1]
    r, c = np.nonzero(mask[:half] == 0)
    src = np.argmax(mask[:half] > 0, axis=0)
    cyl[r, c] = cyl[src[c], c]
    lw = int(cw / 2)
    result = np.zeros((H, cw + 2 * lw, 3), np.uint8)
    result[:, lw:lw + cw] = cyl
    mres = np.zeros((H, cw + 2 * lw), np.uint8)
    mres[:, lw:lw + cw] = mask
    side = cyl[:, 0:lw][:, ::-1]
    result[:, 0:side.shape[1]] = side
    mside = mask[:, :cw // 2][:, ::-1]
    mres[:, :mside.shape[1]] = mside
    side = cyl[:, cw // 2:][:, ::-1]
    result[:, result.shape[1] - side.shape[1]:] = side
    mside = mask[:, cw // 2:][:, ::-1]
    mres[:, mres.shape[1] - side.shape[1]:] = mside
    blur = np.clip(np.rint(gaussian_blur(result.astype(np.float64), 15, 0)), 0, 255).astype(np.uint8)   # :815
    result = np.where(mres[..., None] == 0, blur, result)                                # :816-817
    return result / 255.0


def env_to_xyY(env_bgr):
    """generator.py:407-408."""
    xyY = convert_rgb_to_xyY(env_bgr[..., ::-1])
    xyY[np.isnan(xyY)] = 0
    return xyY
