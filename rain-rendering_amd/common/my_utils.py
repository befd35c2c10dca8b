"""Helpers the hot path needs from the reference's common/my_utils.py (same names,
same argument meaning)."""
import os
import re

import numpy as np


def _natural_key(s):
    return [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', s)]


def os_listdir(path):
    """reference my_utils.os_listdir: natsorted(os.listdir(path)) (my_utils.py:19-20)."""
    return sorted(os.listdir(path), key=_natural_key)


def convert_rgb_to_xyY(array):
    """reference my_utils.convert_rgb_to_xyY (my_utils.py:55-68): row-vector times matrix."""
    mat = np.array([[0.49000, 0.31000, 0.20000], [0.17697, 0.81240, 0.01063], [0.00000, 0.01000, 0.99000]])
    factor = 0.17697
    XYZ = np.dot(array, mat) / factor
    X, Y, Z = XYZ[..., 0], XYZ[..., 1], XYZ[..., 2]
    with np.errstate(divide='ignore', invalid='ignore'):
        x = X / (X + Y + Z)
        y = Y / (X + Y + Z)
    return np.stack([x, y, Y], axis=-1)


def convert_xyY_to_rgb(xyY):
    """reference my_utils.convert_xyY_to_rgb (my_utils.py:71-85)."""
    x, y, Y = xyY[..., 0], xyY[..., 1], xyY[..., 2]
    with np.errstate(divide='ignore', invalid='ignore'):
        X = (Y * x) / y
        Z = (Y * (1 - x - y)) / y
    mat = np.array([[0.41847, -0.15866, -0.082835], [-0.091169, 0.25243, 0.015708], [0.0009209, -0.0025498, 0.1786]])
    return np.dot(np.stack([X, Y, Z], axis=-1), mat)


def crop_center(image, height, width):
    """reference my_utils.crop_center (my_utils.py:88-97)."""
    x1 = int((image.shape[0] - height) / 2)
    y1 = int((image.shape[1] - width) / 2)
    return image[x1:x1 + height, y1:y1 + width]


def particles_path(path, weather):
    """reference my_utils.particles_path (my_utils.py:172)."""
    return os.path.join(path, weather["weather"], "{}mm".format(weather["fallrate"]), '*_camera0.xml')
