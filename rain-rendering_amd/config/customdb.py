"""Template plug-in for a custom dataset (interface of reference config/customdb.py):
<root>/<sequence>/rgb/*.png with depth in <root>/<sequence>/depth/."""
import os

import numpy as np


def resolve_paths(params):
    root = params.images_root
    seqs = [d for d in sorted(os.listdir(root)) if os.path.isdir(os.path.join(root, d, 'rgb'))]
    params.sequences = np.array(seqs)
    assert len(params.sequences) > 0, "There are no valid sequences folder in the dataset root"
    params.images = {s: os.path.join(params.dataset_root, s, 'rgb') for s in params.sequences}
    params.depth = {s: os.path.join(params.depth_root, s, 'depth') for s in params.sequences}
    params.calib = {s: None for s in params.sequences}
    return params


def settings():
    return {
        "cam_hz": 10, "cam_CCD_WH": [1242, 375], "cam_CCD_pixsize": 4.65, "cam_WH": [1242, 375], "cam_focal": 6,
        "cam_gain": 20, "cam_f_number": 6.0, "cam_focus_plane": 6.0, "cam_exposure": 2,
        "cam_pos": [1.5, 1.5, 0.3], "cam_lookat": [1.5, 1.5, -1.], "cam_up": [0., 1., 0.],
        "sequences": {},
    }
