"""Command line of the MI355X rain renderer.  It accepts the flag set of the reference's
main.py (main.py:15-127) and derives the same fields for Generator (main.py:131-161, particle
files main.py:187-220), so an existing invocation keeps working:

    python rain-rendering_amd/main.py --dataset kitti --intensity 25 --frame_end 10
    python -m torch.distributed.run --nproc-per-node 8 rain-rendering_amd/main.py --dataset kitti ...

Missing particle files are produced by this build's own generator (tools/particles.py) where the
reference would start its external, closed-source simulator."""
import argparse
import glob
import os
import sys
import warnings

import numpy as np

if __package__ in (None, ''):
    import importlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    _pkg = importlib.import_module('rain-rendering_amd')
    db = importlib.import_module('rain-rendering_amd.common.db')
    my_utils = importlib.import_module('rain-rendering_amd.common.my_utils')
    Generator = importlib.import_module('rain-rendering_amd.common.generator').Generator
else:
    from .common import db, my_utils
    from .common.generator import Generator

np.random.seed(0)
warnings.filterwarnings("ignore")

_J = os.path.join
# (flags, argparse keywords): names, types and defaults are the reference's; the wording is ours
_FLAGS = [
    (('--dataset',), dict(type=str, required=True, help='dataset name; its data lives in <dataset_root>/<dataset>')),
    (('-k', '--dataset_root'), dict(default=_J('data', 'source'), help='root of the source datasets')),
    (('-p', '--post_fix'), dict(type=str, default='', help='suffix of a GAN-translated dataset variant')),
    (('-s', '--sequences'), dict(default='', help='comma separated sequence prefixes to keep')),
    (('-ns', '--noise_scale'), dict(type=float, default=0.0, help='scale of the angular streak noise')),
    (('-nv', '--noise_std'), dict(type=float, default=0.0, help='standard deviation of the angular streak noise (degrees)')),
    (('-oa', '--opacity_attenuation'), dict(type=float, default=1.0, help='rain layer opacity factor in [0, 1]')),
    (('-r', '--particles'), dict(default=_J('data', 'particles'), help='root of the particle simulations')),
    (('-sd', '--streaks_db'), dict(default=_J('3rdparty', 'rainstreakdb'), help='Garg & Nayar rain streak database')),
    (('-i', '--intensity'), dict(type=str, default='25', help='fall rates in mm/hr, comma separated (e.g. 1,15,25,50)')),
    (('-d', '--depth'), dict(default=_J('data', 'source'), help='root of the depth maps')),
    (('-fs', '--frame_start'), dict(type=int, default=0, help='first frame index')),
    (('-fe', '--frame_end'), dict(type=int, default=None, help='one past the last frame index')),
    (('-fst', '--frame_step'), dict(type=int, default=1, help='frame stride')),
    (('-ff', '--frames'), dict(type=str, default='', help='explicit comma separated frame indices')),
    (('--conflict_strategy',), dict(type=str, default='overwrite', choices=['overwrite', 'skip', 'rename_folder'],
                                    help='what to do when the output folder exists')),
    (('--rendering_strategy',), dict(type=str, default=None, choices=[None, 'white', 'naive_db'], help="None (photometric) or 'white'")),
    (('--output',), dict(default=_J('data', 'output'), help='output root')),
    (('--save_envmap',), dict(action='store_true', help='also write the estimated environment maps')),
    (('--noverbose',), dict(action='store_true', help='no progress output')),
    (('--force_particles',), dict(action='store_true', help='(reference only) re-run the particle simulator')),
    (('--device_particles',), dict(action='store_true', help='simulate the rain particles on the GPU, frame by frame (no particle '
                                                             'file is read or written; this build only)')),
]


def _parse(argv):
    ap = argparse.ArgumentParser(description='Rain rendering on MI355X (hot path in librainhip.so)')
    for names, kw in _FLAGS:
        ap.add_argument(*names, **kw)
    return ap.parse_args(argv)


def _derive(ns):
    """The fields the reference computes after parsing (main.py:131-161)."""
    if ns.force_particles and ns.conflict_strategy == "skip":
        raise AssertionError("If particles simulator is forced, cannot skip")
    ns.verbose = not ns.noverbose
    light_db = _J(ns.streaks_db, 'env_light_database')
    ns.texture = _J(light_db, 'size32')
    ns.norm_coeff = _J(light_db, 'txt', 'normalized_env_max.txt')
    for what, path in (("rainstreakdb database is missing.", ns.streaks_db),
                       ("rainstreakdb database is not valid. Some files are missing.", ns.texture),
                       ("rainstreakdb database is not valid. Some files are missing.", ns.norm_coeff)):
        assert os.path.exists(path), (what, path)
    ns.intensity = [int(v) for v in ns.intensity.split(",")]
    ns.frames = [int(v) for v in ns.frames.split(",")] if ns.frames else ns.frames
    base_name = ns.dataset[:-4] if "_gan" in ns.dataset else ns.dataset
    ns.dataset_root = _J(ns.dataset_root, base_name)
    ns.depth_root = _J(ns.depth, base_name)
    ns.images_root = ns.dataset_root
    ns.calib = None
    assert os.path.exists(ns.images_root), ("Dataset folder does not exist.", ns.images_root)
    wanted = ns.sequences.split(',')
    ns = db.resolve_paths(ns.dataset, ns)                   # dataset plug-in: fills sequences / images / depth / calib
    ns.settings = db.settings(ns.dataset)
    ns.sequences = np.asarray([s for s in ns.sequences if any(s.startswith(w) for w in wanted)])
    ns.weather = np.asarray([dict(weather="rain", fallrate=r) for r in ns.intensity])
    return ns


def _exists(entry):
    if entry is None:
        return True
    return all(os.path.exists(e) for e in entry) if isinstance(entry, list) else os.path.exists(entry)


def _drop_incomplete_sequences(ns):
    """A sequence needs its image folder, depth folder and (if the plug-in names one) calibration."""
    print("\nChecking sequences...")
    print(" {} sequences found: {}".format(len(ns.sequences), list(ns.sequences)))
    for seq in list(ns.sequences):
        problems = [(kind, tree[seq]) for kind, tree in (("images folder", ns.images), ("depth folder", ns.depth),
                                                         ("calib data", ns.calib)) if not _exists(tree[seq])]
        for kind, where in problems:
            print(" Skip sequence '{}': {} is missing {}".format(seq, kind, where))
        if problems:
            ns.sequences = ns.sequences[ns.sequences != seq]
            for tree in (ns.images, ns.depth, ns.calib):
                del tree[seq]
    print("Found {} valid sequence(s): {}".format(len(ns.sequences), list(ns.sequences)))


def _locate_particles(ns):
    """One particle file per (sequence, fall rate) (reference main.py:187-220).  Where the reference launches the external
    weather-particle-simulator for missing (or --force_particles) files, this build runs its own generator
    (tools/particles.py: same settings in, same XML schema out; there is no source of the reference's simulator).

    Under several ranks (torch.distributed.run) rank 0 alone looks, generates and decides; the others receive its list
    (sharding.rank0_decides: a broadcast, which also holds them back until the files are complete).  Every rank
    regenerating the same file, or globbing while a peer rewrites it, would hand the loader a half-written file."""
    if __package__ in (None, ''):
        particles = importlib.import_module('rain-rendering_amd.tools.particles')
        sharding = importlib.import_module('rain-rendering_amd.sharding')
    else:
        from .tools import particles
        from . import sharding
    root = _J(ns.particles, ns.dataset)
    if getattr(ns, 'device_particles', False):
        # BASELINE.json configs[4] from the command line: no XML at all -- the generator's settings go to the GPU with every
        # batch (rr_frame_in.sim) and the drop tables are born there (same model, seed 0, as the files simulate() writes)
        ns.sim_options = {seq: db.sim(ns.dataset, seq, root)["options"] for seq in ns.sequences}
        ns.particles = {seq: [None] * len(ns.weather) for seq in ns.sequences}
        return

    def locate():
        print("\nResolving particles simulations...")
        found = {}
        n_run = 0
        for seq in ns.sequences:
            sim = db.sim(ns.dataset, seq, root)
            found[seq] = []
            for w in ns.weather:
                have = sorted(glob.glob(my_utils.particles_path(sim["path"], w)))
                if ns.force_particles or not have:
                    if n_run == 0:
                        print(" particles simulations to compute...")
                    n_run += 1
                    path = particles.simulate(sim, w, force_recompute=True)      # the file just written, not a re-glob:
                    print("  " + path)                                           # a stale *_camera0.xml may sit beside it
                    found[seq].append(path)
                else:
                    found[seq].append(have[0])
        print(" All particles simulations ready" if n_run == 0 else " All particles simulation completed")
        return found
    ns.particles = sharding.rank0_decides(locate, *sharding.rank_world())


def check_arg(argv):
    ns = _derive(_parse(argv))
    _drop_incomplete_sequences(ns)
    _locate_particles(ns)
    return ns


def main(argv=None):
    print("\nBuilding internal parameters...")
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:      # before anything rank 0 decides for the others (particle files)
        import torch
        import torch.distributed as dist
        # RAIN_DEVICE pins every rank to one device and RAIN_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one
        # GPU): how the GPU test tier runs this driver under N > 1 on the one GPU it has (tests/test_gpu_driver.py)
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get('RAIN_DEVICE', os.environ.get('LOCAL_RANK', '0'))))
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            dist.init_process_group(os.environ.get('RAIN_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo'))
    args = check_arg(sys.argv[1:] if argv is None else argv)
    print("\nRunning renderers...")
    generator = Generator(args)
    generator.run()
    return generator


if __name__ == "__main__":
    main()
