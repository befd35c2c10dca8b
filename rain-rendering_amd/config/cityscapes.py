"""Cityscapes plug-in (interface of reference config/cityscapes.py).  Renders at half
resolution by default (render_scale=2, depth_scale=2) with a 5 ms exposure."""
import glob
import os

import numpy as np


def resolve_paths(params):
    root = params.images_root
    seqs = [d[len(root) + 1:] for d, _, _ in os.walk(root)]
    keep = [s for s in seqs if s and glob.glob(os.path.join(root, s, '*.png')) and 'depth' not in s.split(os.sep)[-2:]]
    params.sequences = np.array(keep)
    assert len(params.sequences) > 0, "There are no valid sequences folder in the dataset root. Have you altered cityscapes file structure ?"
    params.images = {s: os.path.join(params.images_root, s) for s in params.sequences}
    params.depth = {s: os.path.join(params.depth_root, s, os.pardir, 'depth', s.split(os.sep)[-1]) for s in params.sequences}
    params.calib = {s: None for s in params.sequences}
    return params


def settings():
    return {
        "cam_hz": 10, "cam_CCD_WH": [2040, 1016], "cam_CCD_pixsize": 2.2, "cam_WH": [2040, 1016], "cam_focal": 6,
        "cam_gain": 20, "cam_f_number": 6.0, "cam_focus_plane": 6.0, "cam_exposure": 5.0,
        "depth_scale": 2, "render_scale": 2,
        "cam_pos": [1.5, 1.5, 0.3], "cam_lookat": [1.5, 1.5, -1.], "cam_up": [0., 1., 0.],
        "sequences": {"leftImg8bit": {"sim_mode": "steps", "sim_steps": {"cam_motion": np.arange(50., 0. - 1, -1)}}},
    }
