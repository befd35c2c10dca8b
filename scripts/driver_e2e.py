"""End-to-end throughput of the main.py-compatible driver on a synthetic on-disk dataset
(PNG frames + 16-bit depth PNGs + particle XML + streak DB in the reference's layouts):
decode -> pack drops -> rr_pipeline_frames (fog, envmap, streaks) -> encode PNGs.

    python scripts/driver_e2e.py [--frames 48] [--rate 100] [--batch 16]
Prints one JSON line.  Needs a GPU."""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=48)
    ap.add_argument('--rate', type=int, default=100)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument('--height', type=int, default=375)
    ap.add_argument('--width', type=int, default=1242)
    ap.add_argument('--distinct', type=int, default=64, help='distinct image/depth files; the rest of the sequence links to them')
    args = ap.parse_args()
    os.environ['RAIN_BATCH'] = str(args.batch)
    import __graft_entry__ as ge
    ge.build()
    synthetic = importlib.import_module('rain-rendering_amd.synthetic')
    main_mod = importlib.import_module('rain-rendering_amd.main')
    generator_mod = importlib.import_module('rain-rendering_amd.common.generator')
    H, W = args.height, args.width
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, 'source')
        nd = min(args.distinct, args.frames)
        img_dir, dep_dir = synthetic.write_dataset(src, 'kitti', os.path.join('data_object', 'training'), nd, H, W, depth_m=None)
        for i in range(nd, args.frames):               # a long sequence without writing thousands of synthetic PNGs
            os.symlink(os.path.join(img_dir, '%06d.png' % (i % nd)), os.path.join(img_dir, '%06d.png' % i))
            os.symlink(os.path.join(dep_dir, '%06d.png' % (i % nd)), os.path.join(dep_dir, '%06d.png' % i))
        synthetic.write_streak_db(os.path.join(tmp, 'rainstreakdb'))
        frames = synthetic.simulate_particles(4, synthetic.DROPS_PER_RATE[args.rate], W, H)
        xml = os.path.join(tmp, 'particles', 'kitti', 'data_object', 'rain', '%dmm' % args.rate, 'sim_camera0.xml')
        synthetic.write_particles_xml(xml, frames)
        argv = ['--dataset', 'kitti', '-k', src, '-d', src, '-r', os.path.join(tmp, 'particles'), '-sd',
                os.path.join(tmp, 'rainstreakdb'), '-i', str(args.rate), '--output', os.path.join(tmp, 'out'), '--noverbose']
        t0 = time.time()
        gen = main_mod.main(argv)
        t1 = time.time()
        n = len(gen.stats)
        gpu_ms = sum(s['gpu_ms'] for s in gen.stats) / max(n, 1)
        print(json.dumps({"what": "main.py driver end to end (XML load, PNG decode, GPU pipeline, PNG encode)",
                          "frames": n, "seconds": t1 - t0, "frames_per_s": n / (t1 - t0),
                          "pipeline_call_ms_per_frame": gpu_ms, "cores": os.cpu_count(), "cpu_quota": generator_mod._cpu_budget(), "io_threads": gen._io_pool()._max_workers, "timing": gen.timing,
                          "workload": "%dx%d, %d mm/hr" % (W, H, args.rate)}))


if __name__ == '__main__':
    main()
