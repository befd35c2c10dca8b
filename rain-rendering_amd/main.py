"""main.py-compatible command line (same 23 flags as reference main.py:15-127, same derived
fields main.py:131-161, same particles resolution main.py:187-220) driving the MI355X path:

    python -m rain_rendering_amd.main --dataset kitti --intensity 25 --frame_end 10
    python -m torch.distributed.run --nproc-per-node 8 rain-rendering_amd/main.py --dataset kitti ...

The external particle simulator (reference tools/) is NOT driven from here: particle files
must exist (reference main.py would launch AHLSimulation, which has no source in the
reference tree)."""
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


def check_arg(args):
    p = argparse.ArgumentParser(description='Rain renderer method (MI355X hot path)')
    p.add_argument('--dataset', help='Enter dataset name. Dataset data must be located in: DATASET_ROOT/DATASET', type=str, required=True)
    p.add_argument('-k', '--dataset_root', help='Path to database root', default=os.path.join('data', 'source'))
    p.add_argument('-p', '--post_fix', help='Post fix added at the end of the modified gan file', default="", type=str)
    p.add_argument('-s', '--sequences', help='List of sequences comma separated', default='')
    p.add_argument('-ns', '--noise_scale', type=float, default=0.0)
    p.add_argument('-nv', '--noise_std', type=float, default=0.0)
    p.add_argument('-oa', '--opacity_attenuation', help='Opacity attenuation of the rain layer. Values must be between 0 and 1', type=float, default=1.0)
    p.add_argument('-r', '--particles', help='Path to particles simulations', default=os.path.join('data', 'particles'))
    p.add_argument('-sd', '--streaks_db', help='Path to rain streaks database (Garg and Nayar, 2006)', default=os.path.join('3rdparty', 'rainstreakdb'))
    p.add_argument('-i', '--intensity', help='Rain Intensities. List of fall rate comma-separated. E.g.: 1,15,25,50.', type=str, default='25')
    p.add_argument('-d', '--depth', help='Path to depths', default=os.path.join('data', 'source'))
    p.add_argument('-fs', '--frame_start', help='Frame start', type=int, default=0)
    p.add_argument('-fe', '--frame_end', help='Frame end', type=int, default=None)
    p.add_argument('-fst', '--frame_step', help='Frame step', type=int, default=1)
    p.add_argument('-ff', '--frames', type=str, default="")
    p.add_argument('--conflict_strategy', help='Strategy to use if output already exists.', type=str,
                   choices=['overwrite', 'skip', 'rename_folder'], default='overwrite')
    p.add_argument('--rendering_strategy', help='Rendering strategy', choices=[None, 'white', 'naive_db'], type=str, default=None)
    p.add_argument('--output', default=os.path.join('data', 'output'), help='Where to save the output')
    p.add_argument('--save_envmap', help='Save environment maps, useful for debug purposes.', action='store_true')
    p.add_argument('--noverbose', action='store_true')
    p.add_argument('--force_particles', help='Force particles simulator to run even if simulation exist', action='store_true')
    results = p.parse_args(args)

    assert not results.force_particles or results.conflict_strategy != "skip", "If particles simulator is forced, cannot skip"
    results.verbose = not results.noverbose
    results.texture = os.path.join(results.streaks_db, 'env_light_database', 'size32')
    results.norm_coeff = os.path.join(results.streaks_db, 'env_light_database', 'txt', 'normalized_env_max.txt')
    assert os.path.exists(results.streaks_db), ("rainstreakdb database is missing.", results.streaks_db)
    assert os.path.exists(results.texture), ("rainstreakdb database is not valid. Some files are missing.", results.texture)
    assert os.path.exists(results.norm_coeff), ("rainstreakdb database is not valid. Some files are missing.", results.norm_coeff)
    results.intensity = [int(i) for i in results.intensity.split(",")]
    if results.frames:
        results.frames = [int(i) for i in results.frames.split(",")]
    dataset_name = results.dataset if "_gan" not in results.dataset else results.dataset[:-4]
    results.dataset_root = os.path.join(results.dataset_root, dataset_name)
    results.depth_root = os.path.join(results.depth, dataset_name)
    results.calib = None
    results.images_root = os.path.join(results.dataset_root)
    assert os.path.exists(results.images_root), ("Dataset folder does not exist.", results.images_root)
    sequences_filter = results.sequences.split(',')
    results = db.resolve_paths(results.dataset, results)
    results.settings = db.settings(results.dataset)
    results.sequences = np.asarray([seq for seq in results.sequences if np.any([seq[:len(_s)] == _s for _s in sequences_filter])])
    results.weather = np.asarray([{"weather": "rain", "fallrate": i} for i in results.intensity])

    print("\nChecking sequences...")
    print(" {} sequences found: {}".format(len(results.sequences), [s for s in results.sequences]))
    for seq in list(results.sequences):
        valid = True
        if not os.path.exists(results.images[seq]):
            print(" Skip sequence '{}': images folder is missing {}".format(seq, results.images[seq]))
            valid = False
        if not os.path.exists(results.depth[seq]):
            print(" Skip sequence '{}': depth folder is missing {}".format(seq, results.depth[seq]))
            valid = False
        c = results.calib[seq]
        if c is not None and not (np.all([os.path.exists(f) for f in c]) if isinstance(c, list) else os.path.exists(c)):
            print(" Skip sequence '{}': calib data is missing {}".format(seq, c))
            valid = False
        if not valid:
            results.sequences = results.sequences[results.sequences != seq]
            del results.images[seq]
            del results.depth[seq]
            del results.calib[seq]
    print("Found {} valid sequence(s): {}".format(len(results.sequences), [s for s in results.sequences]))

    print("\nResolving particles simulations...")
    particles_root = os.path.join(results.particles, results.dataset)
    sims = {seq: db.sim(results.dataset, seq, particles_root) for seq in results.sequences}
    missing = [(seq, w) for seq in results.sequences for w in results.weather
               if len(glob.glob(my_utils.particles_path(sims[seq]["path"], w))) == 0]
    if missing or results.force_particles:
        raise SystemExit(" {} particles simulations are missing ({}) and the external weather-particle-simulator is not "
                         "driven by this build: generate them with the reference's tools/ or with "
                         "rain_rendering_amd.synthetic".format(len(missing), missing[:3]))
    print(" All particles simulations ready")
    results.particles = {seq: [glob.glob(my_utils.particles_path(sims[seq]["path"], w))[0] for w in results.weather]
                         for seq in results.sequences}
    return results


def main(argv=None):
    print("\nBuilding internal parameters...")
    args = check_arg(sys.argv[1:] if argv is None else argv)
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        import torch
        import torch.distributed as dist
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group(backend)
    print("\nRunning renderers...")
    generator = Generator(args)
    generator.run()
    return generator


if __name__ == "__main__":
    main()
