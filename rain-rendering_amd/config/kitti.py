"""KITTI plug-in (interface of reference config/kitti.py: resolve_paths(params), settings())."""
import os

import numpy as np


def _is_sequence(root, p):
    full = os.path.join(root, p)
    object_like = os.path.isdir(os.path.join(full, 'image_2')) and os.path.isdir(os.path.join(full, 'calib'))
    raw_like = os.path.isdir(os.path.join(full, 'image_02')) and p.endswith('_sync')
    return object_like or raw_like


def resolve_paths(params):
    root = params.images_root
    seqs = [d[len(root) + 1:] for d, _, _ in os.walk(root)]
    params.sequences = np.array([s for s in seqs if s and _is_sequence(root, s)])
    assert len(params.sequences) > 0, "There are no valid sequences folder in the dataset root. Maybe you forgot to download calibration files ?"
    params.images, params.calib, params.depth = {}, {}, {}
    for s in params.sequences:
        if s.startswith('raw_data'):
            params.images[s] = os.path.join(params.dataset_root, s, 'image_02', 'data')
            params.calib[s] = os.path.join(params.dataset_root, s, os.path.pardir, 'calib_cam_to_cam.txt')
        else:
            params.images[s] = os.path.join(params.dataset_root, s, 'image_2')
            cdir = os.path.join(params.dataset_root, s, 'calib')
            params.calib[s] = [os.path.join(cdir, f) for f in os.listdir(cdir) if f.endswith('.txt')]
        params.depth[s] = os.path.join(params.images[s], 'depth')
    return params


def settings():
    return {
        "cam_hz": 10, "cam_CCD_WH": [1242, 375], "cam_CCD_pixsize": 4.65, "cam_WH": [1242, 375], "cam_focal": 6,
        "cam_gain": 20, "cam_f_number": 6.0, "cam_focus_plane": 6.0, "cam_exposure": 2,
        "cam_pos": [1.5, 1.5, 0.3], "cam_lookat": [1.5, 1.5, -1.], "cam_up": [0., 1., 0.],
        "sequences": {"data_object": {"sim_mode": "steps", "sim_steps": {"cam_motion": np.arange(100., 0. - 1, -1)}}},
    }
