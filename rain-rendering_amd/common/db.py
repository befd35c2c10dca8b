"""Settings and dataset plug-in loader with the reference's interface (reference
common/db.py:8-121): settings(db) = defaults merged with config.<db>.settings();
resolve_paths(db, params); sim(db, seq, particles_root).

Plug-ins are looked up first as `config.<name>` on sys.path (so a user's own reference-style
config/<dataset>.py keeps working) and then in this package's config/ directory."""
import importlib
import os
import re

import numpy as np

_settings_defaults = {
    "cam_hz": 10, "cam_CCD_WH": [1242, 375], "cam_CCD_pixsize": 4.65, "cam_WH": [1242, 375], "cam_focal": 6,
    "cam_gain": 20, "cam_f_number": 6.0, "cam_focus_plane": 6.0, "cam_exposure": 2,
    "cam_pos": [1.5, 1.5, 0.3], "cam_lookat": [1.5, 1.5, -1.], "cam_up": [0., 1., 0.],
    "depth_scale": 1, "render_scale": 1,
    "sim_hz": 2000, "sim_mode": "normal", "sim_duration": 34., "sim_steps": {},
    "sequences": {},
}

dbs = {}


def _db(name):
    if name not in dbs:
        try:
            dbs[name] = importlib.import_module("config." + name)
        except ImportError:
            dbs[name] = importlib.import_module(__package__.rsplit('.', 1)[0] + ".config." + name)
    return dbs[name]


def resolve_paths(db, results):
    results = _db(db).resolve_paths(results)
    assert hasattr(results, "images") and hasattr(results, "depth")
    assert hasattr(results, "calib"), "calib files are missing (Kitti format), if no calibration files are provided just set None for each sequence."
    return results


def settings(db):
    s = {**_settings_defaults, **_db(db).settings()}
    s["sequences"] = {re.sub(r'[/|\\]+', os.sep, k): v for k, v in s["sequences"].items()}
    assert s["render_scale"] >= 1 and isinstance(s["render_scale"], int)
    assert s["cam_exposure"] <= 1000. / s["cam_hz"], "Exposure should be lower than 1000./Hz otherwise camera frames temporally overlaps"
    assert s["cam_lookat"][2] < 0, "Z axis should be negative"
    assert np.isclose(np.linalg.norm(s["cam_up"]), 1), "cam_up must be of norm 1"
    return s


def sim(db_s, seq, particles_root):
    db_settings = settings(db_s)
    out = {"path": os.path.join(particles_root, seq), "options": db_settings.copy()}
    match = [s for s in db_settings["sequences"] if re.match(s.replace("\\", "\\\\"), seq) is not None]
    if match:
        out["path"] = os.path.join(particles_root, match[0].replace("*", "x"))
        out["options"] = {**out["options"], **db_settings["sequences"][match[0]]}
        del out["options"]["sequences"]
    else:
        print(" No specific simulation settings found for '{}'. Will fallback to database '{}' settings, if not intentional this might fails.".format(seq, db_s))
    return out
