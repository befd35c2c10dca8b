"""nuScenes plug-in (camera settings of reference config/nuscenes.py:66-81: 5.5 mm, f/1.8, 5 ms, gain 1).

The reference resolves its file lists through the nuScenes devkit (config/nuscenes/nusc_dataset.py: scene tokens ->
CAM_FRONT sample paths) and leaves the sensor size to a preset compiled into its particle simulator ("system code 100",
tools/simulation.py:308-321).  Neither exists here: this plug-in takes the plain folder layout of the custom-database
template (<root>/<scene>/rgb/*.png, <root>/<scene>/depth/*.npy|*.png) and states the CAM_FRONT sensor, 1600 x 900,
explicitly -- what the particle generator (tools/particles.py, csrc/rr_particles.h) simulates on.  The frame index a
nuScenes file maps to is the reference's (generator.py:304-312: the simulated frames spread over the scene's files)."""
import os

import numpy as np


def resolve_paths(params):
    root = params.images_root
    seqs = [d for d in sorted(os.listdir(root)) if os.path.isdir(os.path.join(root, d, 'rgb'))]
    params.sequences = np.array(seqs)
    assert len(params.sequences) > 0, "There are no valid sequences folder in the dataset root."
    params.images = {s: os.path.join(params.dataset_root, s, 'rgb') for s in params.sequences}
    params.depth = {s: os.path.join(params.depth_root, s, 'depth') for s in params.sequences}
    params.calib = {s: None for s in params.sequences}
    return params


def settings():
    return {
        "cam_CCD_WH": [1600, 900], "cam_WH": [1600, 900],
        "cam_focal": 5.5, "cam_gain": 1.0, "cam_f_number": 1.8, "cam_focus_plane": 6.0, "cam_exposure": 5.0,
        "cam_pos": [1.5, 1.5, 0.3], "cam_lookat": [1.5, 1.5, -1.], "cam_up": [0., 1., 0.],
        "sequences": {},
    }
