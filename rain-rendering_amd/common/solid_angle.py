"""Per-pixel solid angles of a lat-long environment map (reference
common/solid_angle.py:5-29,32-45,66-102).  The result depends only on the map's shape,
so it is cached per (He, We)."""
import numpy as np

_cache = {}


def _tetrahedron_solid_angle(a, b, c):
    with np.errstate(invalid='ignore'):
        theta_a = np.arccos(np.sum(b * c, 0))
        theta_b = np.arccos(np.sum(a * c, 0))
        theta_c = np.arccos(np.sum(a * b, 0))
        theta_s = (theta_a + theta_b + theta_c) / 2
        product = np.tan(theta_s / 2) * np.tan((theta_s - theta_a) / 2) * \
            np.tan((theta_s - theta_b) / 2) * np.tan((theta_s - theta_c) / 2)
        product[product < 0] = 0
        return 4 * np.arctan(np.sqrt(product))


def get_solid_angles(img):
    """img: anything with .shape[:2] == (He, We).  Returns float64 (He, We)."""
    h, w = img.shape[0:2]
    key = (h, w)
    if key in _cache:
        return _cache[key]
    u, v = np.meshgrid(np.linspace(0, 1, w + 1), np.linspace(0, 1, h + 1))
    theta = np.pi * (u * 2 - 1)
    phi = np.pi * v
    dx = np.sin(phi) * np.sin(theta)
    dy = np.cos(phi)
    dz = -np.sin(phi) * np.cos(theta)

    def corner(sy, sx):
        return np.vstack((dx[sy, sx].ravel(), dy[sy, sx].ravel(), dz[sy, sx].ravel()))

    lo, hi = slice(None, -1), slice(1, None)
    a, b, c, d = corner(lo, lo), corner(lo, hi), corner(hi, lo), corner(hi, hi)
    omega = _tetrahedron_solid_angle(a, b, c)
    omega += _tetrahedron_solid_angle(b, c, d)
    out = omega.reshape(h, w)
    _cache[key] = out
    return out
