"""TEST INFRASTRUCTURE ONLY -- CPU oracle, never imported by the product path.

Op-for-op numpy restatement of the reference's per-image streak rendering and
compositing path.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package.

Every function cites the reference lines it follows (paths relative to
/root/reference).  Pinning status:

  * PINNED by golden vectors generated from the reference's own code imported in
    the build container (tests/golden/make_golden.py -> tests/golden/*.npz):
    load_streaks_from_xml, classify_drop, take_drop_texture, warping_points,
    compute_circle, compute_fov_plane_points, convert_rgb_to_xyY,
    convert_xyY_to_rgb, get_solid_angles, the colour/blend/accumulate body of
    add_drop_to_image, the streak filter and the epilogue of Generator.run.
  * PARITY UNPINNED (libraries absent, reference has no tests): everything routed
    through oracle/cvlike.py (OpenCV / imutils / pyclipper arithmetic).
  * DELIBERATE, DOCUMENTED DEVIATION: Gaussian defocus weights use det_exp()
    (pure IEEE + - * /, identical bits on every machine and on the GPU) and a
    left-to-right normalisation sum, instead of numpy's platform-dependent SIMD
    exp and pairwise sum inside scipy.ndimage.gaussian_filter.  The blurred tiles
    agree with scipy to <= 4 ulp (tests/test_oracle_pin.py).
"""
import math
import os
import re
from enum import Enum
from xml.etree.ElementTree import parse

import numpy as np

from . import cvlike

# ----------------------------------------------------------------------------
# constants hard-wired in the reference
# ----------------------------------------------------------------------------
FOCUS_PLANE = 6          # generator.py:267
RADIUS = 10              # generator.py:267
FOV_DEG = 165            # generator.py:267
N_FOV = 20               # generator.py:179
SENSOR_PX = 4.65e-06     # bad_weather.py:469
DROP_SIZE = 1.16 * 1e-3  # bad_weather.py:345

ST_OK = 0
ST_FOV_FAIL = 1          # compute_fov_plane_points returned [] (bad_weather.py:698-704)
ST_EMPTY_FOV = 2         # empty intersection -> IndexError (bad_weather.py:372)
ST_BAD_COC = 3           # non-finite circle of confusion (int(10*c) raises)
ST_TOO_BIG = 4           # defocus pad > MAX_SHIFT px: documented limit shared with the HIP library
MAX_SHIFT = 1024


class DropType(Enum):    # bad_weather.py:40-43
    Big = 0
    Medium = 1
    Small = 2


class Streak:            # bad_weather.py:46-60
    def __init__(self):
        self.pid = None
        self.world_position_start = None
        self.world_position_end = None
        self.world_diameter_start = None
        self.world_diameter_end = None
        self.image_position_start = None
        self.image_position_end = None
        self.image_diameter_start = None
        self.image_diameter_end = None
        self.ratio = None
        self.max_width = None
        self.length = None
        self.drop_type = None


class Frame:             # bad_weather.py:66-72
    def __init__(self):
        self.id = None
        self.starting_time = None
        self.exposure_time = None
        self.streaks_count = None
        self.streaks = None


def classify_drop(w):    # bad_weather.py:99-106
    if w >= 4:
        return DropType(0)
    if w > 1:
        return DropType(1)
    return DropType(2)


# ----------------------------------------------------------------------------
# inputs: particles XML and rainstreakdb   
# ----------------------------------------------------------------------------
def load_streaks_from_xml(path, render_scale, image_shape_WH):
    """bad_weather.py:184-241 (pickle cache path omitted: call site passes
    use_pickle=False, generator.py:281; nuscenes_gan branch omitted)."""
    simulation = parse(path).getroot()
    out = {}
    for frame in simulation:
        f = Frame()
        f.id = int(frame.attrib['id'])
        f.exposure_time = int(frame.attrib['t'])
        f.starting_time = int(frame.attrib['d'])
        f.streaks_count = int(frame.attrib['rs'])
        f.streaks = {}
        for drop in frame:
            s = Streak()
            s.pid = int(drop.attrib["pid"])
            s.world_position_start = np.array(drop.attrib["wp1"][1:-1].split(';'), dtype=float)
            s.world_position_end = np.array(drop.attrib["wp2"][1:-1].split(';'), dtype=float)
            s.world_diameter_start = float(drop.attrib['wd1'])
            s.world_diameter_end = float(drop.attrib['wd2'])
            s.image_position_start = np.array(drop.attrib["ip1"][1:-1].split(';'), dtype=float) / render_scale
            s.image_position_end = np.array(drop.attrib["ip2"][1:-1].split(';'), dtype=float) / render_scale
            s.image_diameter_start = float(drop.attrib['iw1']) / render_scale
            s.image_diameter_end = float(drop.attrib['iw2']) / render_scale
            s.image_position_start[1] = image_shape_WH[1] - s.image_position_start[1]
            s.image_position_end[1] = image_shape_WH[1] - s.image_position_end[1]
            s.world_position_start[2] *= -1
            s.world_position_end[2] *= -1
            diff = abs(s.image_position_start - s.image_position_end)
            s.max_width = int(max(s.image_diameter_start, s.image_diameter_end))
            with np.errstate(all='ignore'):
                dir1 = np.array([0, -1])
                dir2 = diff / np.linalg.norm(diff)
                dir2[1] = -dir2[1]
                cos_theta = np.dot(dir1, dir2)
                actual_length = diff[1] / cos_theta
                s.ratio = s.max_width / actual_length
            s.image_position_end = s.image_position_end.round().astype(int)
            s.image_position_start = s.image_position_start.round().astype(int)
            s.length = np.ceil(np.linalg.norm(s.image_position_start - s.image_position_end)).astype(int)
            s.drop_type = classify_drop(s.max_width)
            if s.max_width >= 1 and s.length >= 1:
                f.streaks.update({s.pid: s})
        out.update({f.id: f})
    return out


def _natkey(s):
    return [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', s)]


def load_streak_database(streaks_path, norm_coeff_path):
    """bad_weather.py:108-146.  cv2.imread(IMREAD_ANYDEPTH) -> PIL; natsorted ->
    natural-key sort.  Returns (list of uint8 HxW gray textures, sorted unique ratios)."""
    from PIL import Image
    norm_coeffs = {}
    with open(norm_coeff_path, 'r') as fh:
        lines = fh.readlines()
    coeff = None
    for line in lines:
        if line[:2] == 'cv':
            coeff = int(line[2:])
            continue
        norm_coeffs.update({coeff: [float(v) for v in line.split('\n')[0].split(' ')[:-1]]})
    tmp = []
    ratio = np.array([])
    for file_name in sorted(os.listdir(streaks_path), key=_natkey):
        name = os.path.splitext(file_name)[0]
        coeff, osc = name.split('_')
        coeff = int(coeff[-1:]) if len(coeff) == 3 else int(coeff[-2:])
        osc = int(osc[-1:])
        img = np.array(Image.open(os.path.join(streaks_path, file_name)))
        drop_image_norm = ((255.0 * norm_coeffs[coeff][osc] * img) / 65535.0).astype(np.uint8)
        tmp.append(drop_image_norm)
        ratio = np.append(ratio, tmp[-1].shape[1] / tmp[-1].shape[0])
    return tmp, np.unique(ratio)


def texture_bucket(drop_ratio, ratio):
    """bad_weather.py:250-265: which block of ten textures the randint is drawn from."""
    if drop_ratio < ratio[0]:
        return 0
    if drop_ratio < ratio[1]:
        return 1
    if drop_ratio < ratio[2]:
        return 2
    if drop_ratio < ratio[3]:
        return 3
    return 4


def take_drop_texture_index(drop, ratio):
    """One legacy-RandomState randint per drop, always (bad_weather.py:252-264)."""
    b = texture_bucket(drop.ratio, ratio)
    return np.random.randint(10 * b, 10 * b + 10)


# ----------------------------------------------------------------------------
# colour conversions and solid angles
# ----------------------------------------------------------------------------
def convert_rgb_to_xyY(array):       # my_utils.py:55-68
    mat = np.array([[0.49000, 0.31000, 0.20000], [0.17697, 0.81240, 0.01063], [0.00000, 0.01000, 0.99000]])
    factor = 0.17697
    XYZ = np.dot(array, mat) / factor
    X = XYZ[..., 0]
    Y = XYZ[..., 1]
    Z = XYZ[..., 2]
    with np.errstate(divide='ignore', invalid='ignore'):
        x = X / (X + Y + Z)
        y = Y / (X + Y + Z)
    return np.concatenate([np.expand_dims(x, axis=-1), np.expand_dims(y, axis=-1), np.expand_dims(Y, axis=-1)], axis=-1)


def convert_xyY_to_rgb(xyY):         # my_utils.py:71-85
    x = xyY[..., 0]
    y = xyY[..., 1]
    Y = xyY[..., 2]
    with np.errstate(divide='ignore', invalid='ignore'):
        X = (Y * x) / y
        Z = (Y * (1 - x - y)) / y
    mat = np.array([[0.41847, -0.15866, -0.082835], [-0.091169, 0.25243, 0.015708], [0.0009209, -0.0025498, 0.1786]])
    XYZ = np.concatenate([np.expand_dims(X, axis=-1), np.expand_dims(Y, axis=-1), np.expand_dims(Z, axis=-1)], axis=-1)
    return np.dot(XYZ, mat)


def get_solid_angles(shape_hw):      # solid_angle.py:5-29,32-45,66-102
    h, w = shape_hw
    cols = np.linspace(0, 1, w + 1)
    rows = np.linspace(0, 1, h + 1)
    u, v = np.meshgrid(cols, rows)
    u = u * 2
    theta = np.pi * (u - 1)
    phi = np.pi * v
    dx = np.sin(phi) * np.sin(theta)
    dy = np.cos(phi)
    dz = -np.sin(phi) * np.cos(theta)
    a = np.vstack((dx[:-1, :-1].ravel(), dy[:-1, :-1].ravel(), dz[:-1, :-1].ravel()))
    b = np.vstack((dx[:-1, 1:].ravel(), dy[:-1, 1:].ravel(), dz[:-1, 1:].ravel()))
    c = np.vstack((dx[1:, :-1].ravel(), dy[1:, :-1].ravel(), dz[1:, :-1].ravel()))
    d = np.vstack((dx[1:, 1:].ravel(), dy[1:, 1:].ravel(), dz[1:, 1:].ravel()))

    def tet(a, b, c):
        with np.errstate(invalid='ignore'):
            ta = np.arccos(np.sum(b * c, 0))
            tb = np.arccos(np.sum(a * c, 0))
            tc = np.arccos(np.sum(a * b, 0))
            ts = (ta + tb + tc) / 2
            product = np.tan(ts / 2) * np.tan((ts - ta) / 2) * np.tan((ts - tb) / 2) * np.tan((ts - tc) / 2)
            product[product < 0] = 0
            return 4 * np.arctan(np.sqrt(product))

    omega = tet(a, b, c)
    omega += tet(b, c, d)
    return omega.reshape(h, w)


# ----------------------------------------------------------------------------
# deterministic exp and the defocus blur   
# ----------------------------------------------------------------------------
_LN2_HI = 6.93147180369123816490e-01
_LN2_LO = 1.90821492927058770002e-10
_INV_LN2 = 1.44269504088896338700e+00
_EXP_C = [1.0 / math.factorial(n) for n in range(14)]


def det_exp(x):
    """exp(x) for -700 < x <= 0 from + - * / only (no FMA), so that numpy, g++ and
    the gfx950 kernels produce identical bits.  k = rint(x/ln2); r = x - k*ln2 (two
    pieces); degree-13 Taylor polynomial by Horner; ldexp.  <= 1 ulp."""
    x = np.asarray(x, np.float64)
    k = np.rint(x * _INV_LN2)
    r = (x - k * _LN2_HI) - k * _LN2_LO
    p = np.full_like(r, _EXP_C[13])
    for n in range(12, -1, -1):
        p = p * r + _EXP_C[n]
    return np.ldexp(p, k.astype(np.int64))


def gaussian_weights(sigma):
    """scipy.ndimage._gaussian_kernel1d(sigma, 0, int(4*sigma+0.5)) with det_exp and a
    sequential normalisation sum (see module docstring)."""
    radius = int(4.0 * float(sigma) + 0.5)
    sigma2 = sigma * sigma
    x = np.arange(-radius, radius + 1)
    phi = det_exp(-0.5 / sigma2 * x ** 2)
    tot = 0.0
    for v in phi:
        tot = tot + v
    return phi / tot, radius


def correlate1d_sym(a, w, r, axis):
    """scipy ni_filters.c NI_Correlate1D, symmetric-kernel branch, zero extension
    (the tile is zero-padded by >= radius beforehand, so mode='reflect' never sees
    data).  tmp = x[l]*w[r]; for ii=-r..-1: tmp += (x[l+ii] + x[l-ii])*w[ii+r]."""
    a = np.moveaxis(a, axis, 0)
    n = a.shape[0]
    z = np.zeros((r,) + a.shape[1:])
    p = np.concatenate([z, a, z], axis=0)
    out = p[r:r + n] * w[r]
    for ii in range(-r, 0):
        out = out + (p[r + ii:r + ii + n] + p[r - ii:r - ii + n]) * w[ii + r]
    return np.moveaxis(out, 0, axis)


def gaussian_filter_2d(tile, sigma1, sigma2):
    """scipy.ndimage.gaussian_filter(tile, [sigma1, sigma2, 0]) (bad_weather.py:296)
    for a (h, w, C) tile: axis 0 then axis 1; an axis with sigma <= 1e-15 is skipped."""
    out = tile
    for axis, sigma in ((0, sigma1), (1, sigma2)):
        if sigma > 1e-15:
            w, r = gaussian_weights(sigma)
            out = correlate1d_sym(out, w, r, axis)
    return out


def compute_circle(o, f, N, focus_plane=FOCUS_PLANE):     # bad_weather.py:464-469
    result = ((o - focus_plane) * f ** 2) / (o * (focus_plane - f) * N)
    return result / SENSOR_PX


def circle_of_confusion(drop, drop_distance, f, N):       # bad_weather.py:286-298
    with np.errstate(all='ignore'):
        c = abs(compute_circle(abs(drop_distance), f, N))
    sigma1, sigma2 = c, c / 2
    shift = int(10 * c)      # raises on nan/inf -> drop skipped (generator.py:185)
    drop2 = np.pad(drop, ((shift, shift), (shift, shift), (0, 0)), mode='constant')   # cv2.copyMakeBorder
    drop2 = gaussian_filter_2d(drop2, sigma1, sigma2)
    return drop2, shift


# ----------------------------------------------------------------------------
# per-drop geometry
# ----------------------------------------------------------------------------
def warping_points(drop, tex_shape, image_width, image_height):   # bad_weather.py:300-329
    x0 = round(drop.image_position_start[0])
    x1 = round(drop.image_position_end[0])
    y0 = round(drop.image_position_start[1])
    y1 = round(drop.image_position_end[1])
    d0 = np.floor(drop.image_diameter_start)
    d1 = np.floor(drop.image_diameter_end)
    minx = max(min(x0, x1), 0)
    miny = max(min(y0, y1), 0)
    maxx = min(max(x0 + d0, x1 + d1), image_width)
    maxy = min(max(y0, y1), image_height)
    epsilon = 0.001
    p1 = np.float32([[0, 0], [tex_shape[1], 0], [tex_shape[1], tex_shape[0]], [0, tex_shape[0]]])
    p2 = np.float32([[x0 - minx, y0 - miny],
                     [x0 - minx + d0, y0 - miny],
                     [x1 - minx + d1 + epsilon, y1 - miny],
                     [x1 - minx + epsilon, y1 - miny]])
    return p1, p2, np.array([maxx, maxy]), np.array([minx, miny])


def _normalize(v):
    return v / np.linalg.norm(v)


def _rotation_matrix(axis, theta):                         # bad_weather.py:532-538
    axis = np.asarray(axis)
    c, s = np.cos(theta), np.sin(theta)
    skv = np.roll(np.roll(np.diag(axis.flatten()), 1, 1), -1, 0)
    return (c * np.identity(3)) + s * (skv - skv.T) + ((1 - c) * np.outer(axis, axis))


def _intersection_sphere(position, direction, radius):     # bad_weather.py:540-568
    dx, dy, dz = direction
    x0, y0, z0 = position
    R = radius
    cx = cy = cz = 0
    a = dx * dx + dy * dy + dz * dz
    b = 2 * dx * (x0 - cx) + 2 * dy * (y0 - cy) + 2 * dz * (z0 - cz)
    c = cx * cx + cy * cy + cz * cz + x0 * x0 + y0 * y0 + z0 * z0 + -2 * (cx * x0 + cy * y0 + cz * z0) - R * R
    disc = b ** 2 - 4 * a * c
    sqrt_disc = np.sqrt(disc)
    t1 = (-b + sqrt_disc) / (2 * a)
    return position + (t1 * direction)


def _cart2sph(p):                                          # bad_weather.py:570-586
    x, y, z = p
    r = np.sqrt(x ** 2 + y ** 2 + z ** 2)
    el = np.arctan2(z, np.sqrt(x ** 2 + y ** 2))
    az = np.arctan2(y, x)
    if az < 0:
        az += 2 * np.pi
    if el < 0:
        el += 2 * np.pi
    if az > np.pi * 2:
        az -= 2 * np.pi
    if el > np.pi * 2:
        el -= 2 * np.pi
    return az, el, r


def compute_fov_plane_points(wps, wpe, radius, fov, N, env_shape):
    """bad_weather.py:596-704.  Returns the (20|24, 2) float vertex array, or an empty
    array where the reference's bare `except:` fires."""
    camera = np.array([0, 0, 0])
    try:
        with np.errstate(all='ignore'):
            drop_position = np.array((wps + wpe) / 2)
            drop_position[1], drop_position[2] = drop_position[2], drop_position[1].copy()
            drop_direction = _normalize(drop_position - camera)
            theta = np.deg2rad(fov / 2)
            a = drop_direction[0]
            b = drop_direction[1]
            c = drop_direction[2]
            d = np.dot(drop_position, drop_direction)
            if b == 0:
                b = 0.001
            px = drop_position[1]
            pz = 0
            py = (-a * px + d - c * pz) / b
            point = np.array([px, py, pz])
            u = _normalize(drop_position - point)
            assert (np.all(~np.isnan(u)) and "Some values are NAN")
            rot_vec = np.cross(u, drop_direction)
            rot_mat = _rotation_matrix(rot_vec, -theta)
            v = np.dot(drop_direction, rot_mat)
            phi = np.arange(0, 2 * np.pi, (2 * np.pi) / N)
            vectors = np.array([])
            for angle in phi:
                M = _rotation_matrix(drop_direction, angle)
                vectors = np.append(vectors, [np.dot(v, M)])
            vectors = np.reshape(vectors, (-1, 3))
            points = np.array([])
            for dir_v in vectors:
                points = np.append(points, _intersection_sphere(drop_position, dir_v, radius))
            points = np.reshape(points, (-1, 3))
            azs = np.array([])
            points_image = np.array([])
            for p in points:
                azimuth, elevation, r = _cart2sph(p)
                azimuth = ((2 * np.pi - azimuth) - np.pi / 2)
                azimuth = azimuth % (2 * np.pi)
                u = azimuth / (2 * np.pi)
                elevation = (elevation + np.pi / 2)
                elevation = elevation % (2 * np.pi)
                v = 1. - elevation / np.pi
                azs = np.append(azs, azimuth)
                points_image = np.append(points_image, [u * env_shape[1], v * env_shape[0]])
            points_image = np.reshape(points_image, (-1, 2))
            azs = np.append(azs, azs[0])
            cond = np.bitwise_or(np.isclose(np.diff(azs), 0), np.diff(azs) < 0)
            cond_true = cond
            cond_false = ~cond
            count_true = np.sum(cond_true)
            count_false = np.sum(cond_false)
            pos_true = np.where(cond_true)[0][0]
            pos_false = np.where(cond_false)[0][0]
            rows, cols = env_shape[:2]
            if count_true == 1:      # top
                final_pts = np.vstack([points_image[:pos_true + 1],
                                       [cols, points_image[pos_true][1]],
                                       [cols, 0],
                                       [0, 0],
                                       [0, points_image[np.mod(pos_true + 1, N)][1]],
                                       points_image[pos_true + 1:]])
            elif count_false == 1:   # bottom
                final_pts = np.vstack([points_image[:pos_false + 1],
                                       [0, points_image[pos_false][1]],
                                       [0, rows],
                                       [cols, rows],
                                       [cols, points_image[np.mod(pos_false + 1, N)][1]],
                                       points_image[pos_false + 1:]])
            else:
                final_pts = points_image
            return np.array(final_pts)
    except Exception:
        return np.array([])


# ----------------------------------------------------------------------------
# one drop: texture -> tile   (generator.py:119-174)
# ----------------------------------------------------------------------------
def make_drop_tile(drop, tex_u8, noise_deg, W, H, rot=None):
    """Returns (tile HxWx4 f64, minC int[2]).  Mutates drop.image_position_* exactly as
    generator.py:152-161 does for non-Big drops.  rot = (cos, sin) of -(theta + noise) * pi / 180: take the rotation
    from a drop RECORD (render_drop_records) instead of evaluating generator.py:138-145 here."""
    tex = tex_u8.astype(np.float64) / 255.0          # bad_weather.py:252 (gray; 3 identical channels)
    if drop.drop_type == DropType.Big:
        pts1, pts2, maxC, minC = warping_points(drop, tex.shape, W, H)
        shape = np.subtract(maxC, minC).astype(int)
        M = cvlike.get_perspective_transform(pts1, pts2)
        g = cvlike.warp_perspective_cubic(tex, M, max(int(shape[0]), 1), max(int(shape[1]), 1))
        g = np.clip(g, 0, 1)
    else:
        noise = noise_deg
        dir1 = drop.image_position_start - drop.image_position_end
        n1 = np.linalg.norm(dir1)
        dir1 = dir1 / n1
        dir2 = np.array([0, -1])
        theta = np.rad2deg(np.arccos(np.dot(dir1, dir2)))
        nx, ny = np.cos(np.deg2rad(noise)), np.sin(np.deg2rad(noise))
        mean_x = (drop.image_position_end[0] + drop.image_position_start[0]) / 2
        mean_y = (drop.image_position_end[1] + drop.image_position_start[1]) / 2
        drop.image_position_start[:] = \
            (drop.image_position_start[0] - mean_x) * nx - (drop.image_position_start[1] - mean_y) * ny + mean_x, \
            (drop.image_position_start[0] - mean_x) * ny + (drop.image_position_start[1] - mean_y) * nx + mean_y
        drop.image_position_end[:] = \
            (drop.image_position_end[0] - mean_x) * nx - (drop.image_position_end[1] - mean_y) * ny + mean_x, \
            (drop.image_position_end[0] - mean_x) * ny + (drop.image_position_end[1] - mean_y) * nx + mean_y
        ang = -(theta + noise) * (np.pi / 180)        # getRotationMatrix2D(center, -angle, 1)
        g = cvlike.rotate_bound(tex, np.cos(ang), np.sin(ang)) if rot is None else cvlike.rotate_bound(tex, rot[0], rot[1])
        if drop.image_position_end[0] > W // 2:
            g = cvlike.flip0(g)
        height = max(abs(drop.image_position_end[1] - drop.image_position_start[1]), 2)
        width = max(abs(drop.image_position_end[0] - drop.image_position_start[0]), drop.max_width + 2)
        g = cvlike.resize_area(g, int(width), int(height))
        g = np.clip(g, 0, 1)
        minC = drop.image_position_start
    tile = np.dstack([g, g, g, g])                    # generator.py:174 (alpha = channel 0)
    return tile, np.array(minC)


# ----------------------------------------------------------------------------
# one drop: colour, defocus, placement, blend     (bad_weather.py:336-462)
# ----------------------------------------------------------------------------
class FrameConsts:
    """Per-frame reductions that add_drop_to_image recomputes for every drop
    (bad_weather.py:403-407); hoisting them does not change any bit."""

    def __init__(self, env_map_xyY, solid_angle_map):
        self.sum_omega = np.sum(solid_angle_map)
        ambient_lum = env_map_xyY[..., 2] * solid_angle_map
        self.ambient_lum = np.sum(ambient_lum) / self.sum_omega


def fov_colour(env_map_xyY, solid_angle_map, poly_int, fc, faithful=True):
    """bad_weather.py:383-409: (fov_xy_avg[2], drop_Y) or None if the mask is empty."""
    rows, cols = env_map_xyY.shape[:2]
    if faithful:
        mask = np.zeros((rows, cols), np.float64)
        cvlike.fill_fov_mask(mask, poly_int)
        mask_env = mask.astype(bool)
        if not mask_env.any():
            return None
        fov_solid_angle = solid_angle_map[mask_env].copy()
        fov_envmap = env_map_xyY[mask_env].copy()
        fov_xyY = (fov_envmap * np.expand_dims(fov_solid_angle, axis=-1)).sum(axis=0)
        s_omega = np.sum(fov_solid_angle)
    else:
        # same spans, summed row by row (used only for large test cases: colour is
        # compared with a tolerance)
        y0, xl, xr = cvlike.fov_rowspans(poly_int, rows, cols)
        fov_xyY = np.zeros(3)
        s_omega = 0.0
        any_px = False
        for k in range(len(xl)):
            if xl[k] <= xr[k]:
                any_px = True
                om = solid_angle_map[y0 + k, xl[k]:xr[k] + 1]
                fov_xyY = fov_xyY + (env_map_xyY[y0 + k, xl[k]:xr[k] + 1] * om[:, None]).sum(axis=0)
                s_omega = s_omega + om.sum()
        if not any_px:
            return None
    fov_xy_avg = fov_xyY[:2] / s_omega
    avg_fov_lum = fov_xyY[2] / fc.sum_omega
    drop_Y = 0.94 * avg_fov_lum + 0.06 * fc.ambient_lum
    return fov_xy_avg, drop_Y


def _visible(drop, scene_depth, y0, x0, shape_hw):
    """DEPTH-OCCLUSION OPTION (not in the reference's output; the reference only sketches a depth test in the dead
    common/drop_depth_map.py behind USE_DEPTH_WEIGHTING = 0, generator.py:20,339-341).  Definition, shared with
    tests/hostemu and the library's RR_OPT_DEPTH_OCCLUSION: a drop is HIDDEN at a pixel -- neither blended nor added to
    the mask there -- iff its distance from the camera |world_position_start.z| is greater than the scene depth (metres)
    at that pixel; a NaN depth hides nothing.  Returns the boolean "visible" map of the region the tile covers, or None
    when no depth buffer is given (the reference's behaviour)."""
    if scene_depth is None:
        return None
    region = scene_depth[y0:y0 + shape_hw[0], x0:x0 + shape_hw[1]].astype(np.float64)
    return ~(abs(float(drop.world_position_start[2])) > region)


def add_drop_to_image(env_map_xyY, solid_angle_map, fc, drop_fov_pts, drop_minC, bg_shape, rainy_bg, rainy_mask,
                      tile, drop, cam, opacity_attenuation=1.0, faithful=True, rendering_strategy=None, scene_depth=None):
    """bad_weather.py:336-462: the default rendering strategy and 'white' ('naive_db' reads a
    non-existent attribute in the reference, bad_weather.py:355, and cannot run).  Raises (like
    the reference) when the drop must be skipped; the caller turns that into a status.  Returns the reference's
    (drop_vis, drop_blend, drop_minC) (:462).  scene_depth (H x W metres): the depth-occlusion option, see _visible."""
    exposure_time = cam['exposure_ms'] / 1000.
    if rendering_strategy in ['white']:
        # bad_weather.py:349-353: gray tile, no colour, no defocus, no clamp of the origin
        tau_zero = np.sqrt(DROP_SIZE) / 50
        length_opacity = 1.
        tau_one = exposure_time * length_opacity
        drop_minC = np.array(drop_minC)
        y0, x0 = int(drop_minC[1]), int(drop_minC[0])
        rainy_bg_occ = rainy_bg[y0:y0 + tile.shape[0], x0:x0 + tile.shape[1], :].copy()
        rainy_mask_occ = rainy_mask[y0:y0 + tile.shape[0], x0:x0 + tile.shape[1]].copy()
        drop_vis = tile[:rainy_bg_occ.shape[0], :rainy_bg_occ.shape[1]]
        drop_vis_alpha = drop_vis[:, :, 3]
        drop_vis_alpha_ = np.expand_dims(drop_vis_alpha, axis=-1)
        rainy_bg_occ = ((1. - ((drop_vis_alpha_ * tau_one) / exposure_time)) * rainy_bg_occ) + drop_vis[:, :, :3] * (
            tau_one / tau_zero)
        rainy_bg_occ = np.clip(rainy_bg_occ, 0, 1)
        vis = _visible(drop, scene_depth, y0, x0, tile.shape[:2])
        if vis is not None:
            rainy_bg_occ = np.where(vis[..., None], rainy_bg_occ, rainy_bg[y0:y0 + tile.shape[0], x0:x0 + tile.shape[1], :])
            drop_vis_alpha = np.where(vis, drop_vis_alpha, 0.0)
        rainy_mask_occ += drop_vis_alpha
        rainy_bg[y0:y0 + rainy_bg_occ.shape[0], x0:x0 + rainy_bg_occ.shape[1]] = rainy_bg_occ
        rainy_mask[y0:y0 + tile.shape[0], x0:x0 + tile.shape[1]] = rainy_mask_occ
        return drop_vis, rainy_bg_occ, drop_minC
    if len(drop_fov_pts) == 0:
        raise IndexError(ST_FOV_FAIL)                # pyclipper.AddPath on an empty path
    if not np.all(np.isfinite(drop_fov_pts)):
        raise IndexError(ST_FOV_FAIL)                # Clipper range error on NaN coordinates
    poly_int = cvlike.polygon_to_int(drop_fov_pts)
    if cvlike.polygon_all_collinear(poly_int):
        raise IndexError(ST_FOV_FAIL)                # ClipperException: AddPath rejects a path without three non-collinear vertices
    d_avg = (drop.image_diameter_start + drop.image_diameter_end) / 2.

    drop_xyY = convert_rgb_to_xyY(tile[..., :3])
    drop_xyY[np.isnan(drop_xyY)] = 0
    col = fov_colour(env_map_xyY, solid_angle_map, poly_int, fc, faithful)
    if col is None:
        raise IndexError(ST_EMPTY_FOV)               # solution[0] on an empty intersection
    fov_xy_avg, drop_Y = col
    drop_xyY_fov_color = drop_xyY.copy()
    drop_xyY_fov_color[..., :2] = fov_xy_avg
    drop_xyY_fov_color[..., 2] *= drop_Y
    drop_color_rgb = convert_xyY_to_rgb(drop_xyY_fov_color)
    drop_color_bgr = drop_color_rgb[..., ::-1]
    tile = tile.copy()
    tile[..., :3][tile[..., 3] > 0] = drop_color_bgr[tile[..., 3] > 0]

    with np.errstate(all='ignore'):
        c_probe = abs(compute_circle(abs(drop.world_position_start[2]), cam['focal_m'], cam['f_number']))
    if not np.isfinite(c_probe):
        raise IndexError(ST_BAD_COC)                 # int(10*c) raises ValueError/OverflowError
    if 10 * c_probe >= MAX_SHIFT + 1:
        raise IndexError(ST_TOO_BIG)                 # NOT in the reference: see DESIGN.md "limits"
    tile, shift = circle_of_confusion(tile, drop.world_position_start[2], cam['focal_m'], cam['f_number'])

    H, W = bg_shape[:2]
    drop_minC_tmp = drop_minC - shift
    drop_minC = np.array([np.clip(drop_minC_tmp[0], 0, W), np.clip(drop_minC_tmp[1], 0, H)])
    delta = drop_minC - drop_minC_tmp
    tile = tile[:delta[1]] if delta[1] < 0 else tile[delta[1]:]
    tile = tile[:, :delta[0]] if delta[0] < 0 else tile[:, delta[0]:]

    tau_zero = np.sqrt(DROP_SIZE) / 50
    length_opacity = opacity_attenuation * d_avg / (drop.length + d_avg)
    tau_one = exposure_time * length_opacity

    y0, x0 = int(drop_minC[1]), int(drop_minC[0])
    rainy_bg_occ = rainy_bg[y0:y0 + tile.shape[0], x0:x0 + tile.shape[1], :].copy()
    rainy_mask_occ = rainy_mask[y0:y0 + tile.shape[0], x0:x0 + tile.shape[1]].copy()
    drop_vis = tile[:rainy_bg_occ.shape[0], :rainy_bg_occ.shape[1]]
    drop_vis_alpha = drop_vis[:, :, 3]
    drop_vis_alpha_ = np.expand_dims(drop_vis_alpha, axis=-1)
    rainy_bg_occ = ((1. - ((drop_vis_alpha_ * tau_one) / exposure_time)) * rainy_bg_occ) + drop_vis[:, :, :3] * (
        tau_one / tau_zero)
    rainy_bg_occ = np.clip(rainy_bg_occ, 0, 1)
    vis = _visible(drop, scene_depth, y0, x0, tile.shape[:2])
    if vis is not None:
        rainy_bg_occ = np.where(vis[..., None], rainy_bg_occ, rainy_bg[y0:y0 + tile.shape[0], x0:x0 + tile.shape[1], :])
        drop_vis_alpha = np.where(vis, drop_vis_alpha, 0.0)
    rainy_mask_occ += drop_vis_alpha
    rainy_bg[y0:y0 + rainy_bg_occ.shape[0], x0:x0 + rainy_bg_occ.shape[1]] = rainy_bg_occ
    rainy_mask[y0:y0 + tile.shape[0], x0:x0 + tile.shape[1]] = rainy_mask_occ
    return drop_vis, rainy_bg_occ, drop_minC


# ----------------------------------------------------------------------------
# one frame     (generator.py:389-467)
# ----------------------------------------------------------------------------
def streak_filter(streaks, imW, imH):                # generator.py:413-420
    return {k: v for k, v in streaks.items() if
            1 <= v.max_width < max(imH, imW) and
            1 <= v.length < max(imH, imW) and
            ((0 <= v.image_position_start[0] < imW and 0 <= v.image_position_start[1] < imH) or
             (0 <= v.image_position_end[0] < imW and 0 <= v.image_position_end[1] < imH))}


def quantise_image(rainy_bg, bg):
    """generator.py:461-466 + matplotlib's float->uint8 rule (truncation): returns the
    RGB uint8 image plt.imsave would write (alpha channel omitted)."""
    difference_mean = np.mean(rainy_bg) - np.mean(bg)
    out = np.clip((rainy_bg - difference_mean)[..., ::-1], 0, 1)
    return (out * 255).astype(np.uint8)


def quantise_mask(rainy_mask):
    """Decision D1: int32 export of the float64 accumulator."""
    return np.floor(rainy_mask * 255).astype(np.int32)


def render_frame(bg, rainy_bg, env_map_xyY, solid_angle_map, streak_list, textures, ratio, cam,
                 frame_seed, noise_std=0.0, noise_scale=0.0, opacity_attenuation=1.0,
                 faithful=True, max_drops=None, rendering_strategy=None, first_drop=0, scene_depth=None):
    """The hot loop of Generator.run for one frame (generator.py:318,389-394,428-438,461-467).

    streak_list: the already filtered list of Streak objects (mutated like the reference does).
    first_drop > 0 renders the window [first_drop, max_drops) only: the earlier drops still consume their
    random draws (so the window sees the reference's RNG stream) but are not composited (test windows;
    noise-free scenes only, since the skipped drops' in-place end-point rotation is not replayed).
    scene_depth (H x W, metres): the depth-occlusion OPTION (default None = the reference's output), see _visible.
    Returns dict(rainy_bg f64, mask f64, mask_i32, image_u8 RGB, status int32[n])."""
    np.random.seed(frame_seed)                        # generator.py:318
    H, W = bg.shape[:2]
    rainy_bg = rainy_bg.copy()
    rainy_mask = np.zeros((H, W), np.float64)
    fc = FrameConsts(env_map_xyY, solid_angle_map)
    n = len(streak_list) if max_drops is None else min(max_drops, len(streak_list))
    status = np.zeros(n, np.int32)
    for i in range(n):
        drop = streak_list[i]
        tex_idx = take_drop_texture_index(drop, ratio)                    # RNG draw 1 (always)
        noise = 0.0
        if drop.drop_type != DropType.Big:
            noise = np.random.normal(0.0, noise_std) * noise_scale       # RNG draw 2 (generator.py:136)
        if i < first_drop:
            continue
        tile, minC = make_drop_tile(drop, textures[tex_idx], noise, W, H)
        pts = compute_fov_plane_points(drop.world_position_start, drop.world_position_end,
                                       RADIUS, FOV_DEG, N_FOV, env_map_xyY.shape)
        try:
            add_drop_to_image(env_map_xyY, solid_angle_map, fc, pts, minC, bg.shape, rainy_bg, rainy_mask,
                              tile, drop, cam, opacity_attenuation, faithful, rendering_strategy, scene_depth)
        except IndexError as e:                       # generator.py:185-189: any exception == skip
            status[i] = e.args[0] if e.args and isinstance(e.args[0], int) else ST_FOV_FAIL
    return dict(rainy_bg=rainy_bg, mask=rainy_mask, mask_i32=quantise_mask(rainy_mask),
                image_u8=quantise_image(rainy_bg, bg), status=status[first_drop:])


def render_drop_records(bg, rainy_bg, env_map_xyY, solid_angle_map, records, textures, cam, opacity_attenuation=1.0,
                        faithful=True, rendering_strategy=None, scene_depth=None):
    """render_frame for a drop table given as rr_drop RECORDS (a numpy structured array with the fields of
    include/rainhip.h rr_drop: what the product's host packer -- or the library's device-side particle generator --
    hands to the kernels): the reference's per-drop path with the random draws already made (tex_index) and the streak
    rotation taken from the record (rot_cos / rot_sin; no angular noise).  Everything else -- tile synthesis, field of
    view, colour, defocus, placement, blend, epilogue -- is render_frame's."""
    H, W = bg.shape[:2]
    rainy_bg = rainy_bg.copy()
    rainy_mask = np.zeros((H, W), np.float64)
    fc = FrameConsts(env_map_xyY, solid_angle_map)
    status = np.zeros(len(records), np.int32)
    for i, r in enumerate(records):
        drop = Streak()
        drop.pid = i
        drop.world_position_start = np.array(r['wps'], np.float64)
        drop.world_position_end = np.array(r['wpe'], np.float64)
        drop.image_position_start = np.array([int(r['x0']), int(r['y0'])])
        drop.image_position_end = np.array([int(r['x1']), int(r['y1'])])
        drop.image_diameter_start, drop.image_diameter_end = float(r['iw1']), float(r['iw2'])
        drop.max_width, drop.length = int(r['max_width']), int(r['length'])
        drop.drop_type = DropType(int(r['type']))
        tile, minC = make_drop_tile(drop, textures[int(r['tex_index'])], 0.0, W, H, rot=(float(r['rot_cos']), float(r['rot_sin'])))
        pts = compute_fov_plane_points(drop.world_position_start, drop.world_position_end, RADIUS, FOV_DEG, N_FOV, env_map_xyY.shape)
        try:
            add_drop_to_image(env_map_xyY, solid_angle_map, fc, pts, minC, bg.shape, rainy_bg, rainy_mask,
                              tile, drop, cam, opacity_attenuation, faithful, rendering_strategy, scene_depth)
        except IndexError as e:
            status[i] = e.args[0] if e.args and isinstance(e.args[0], int) else ST_FOV_FAIL
    return dict(rainy_bg=rainy_bg, mask=rainy_mask, mask_i32=quantise_mask(rainy_mask),
                image_u8=quantise_image(rainy_bg, bg), status=status)
