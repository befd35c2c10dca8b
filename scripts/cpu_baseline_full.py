"""The CPU baselines of SURVEY 8(d) in full: WHOLE frames (bench.py times a bounded sample), on the host cores of the box
this runs on.  No GPU needed.

  (i)   1 process, 1 core: the numpy oracle in its op-for-op mode (per-drop masked reduction over the whole environment
        map, like the reference), >= 3 frames per configuration, median seconds per frame and ms per drop;
  (ii)  P = min(10, cores) processes (reference main_threaded.py:176), aggregate frames/s;
  (iii) the g++ -O2 build of the kernel arithmetic (tests/hostemu), 1 core: the stronger baseline.

    python scripts/cpu_baseline_full.py [--frames 3] [--rates 25,100] > profiles/rNN_cpu_baseline.json
The frames of (i) run concurrently, one single-threaded process each (they do not share cores as long as the box has
enough of them; the quota and the core count are recorded)."""
import argparse
import json
import multiprocessing as mp
import os
import sys
import tempfile
import time

os.environ.setdefault('OMP_NUM_THREADS', '1')
os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def _scene(tmp, rate, n_frames):
    import helpers as h
    return h, h.Scene(tmp, 375, 1242, h.synthetic.DROPS_PER_RATE[rate], n_frames=n_frames, seed0=3000)


def _one_frame(args):
    tmp, rate, n_frames, i, what = args
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):
        h, sc = _scene(tmp, rate, n_frames)
        bg, env = sc.frame_inputs(i)
    if what == 'oracle':
        t0 = time.perf_counter()
        out = h.oracle_render(sc, i, bg, bg, env, faithful=True)
        return time.perf_counter() - t0, len(out['status'])
    drops = sc.product_drops(i)
    h.hostemu()
    t0 = time.perf_counter()
    h.emu_render(sc, bg, bg, env, drops)
    return time.perf_counter() - t0, len(drops)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=3)
    ap.add_argument('--rates', default='25,100')
    a = ap.parse_args()
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        budget = int(int(quota) / int(period)) if quota != 'max' else os.cpu_count()
    except (OSError, ValueError):
        budget = os.cpu_count()
    model = next((l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')), None)
    res = {"cpu_model": model, "nproc": os.cpu_count(), "cpu_quota": budget, "frame": "1242x375 (KITTI shape), environment map 1909x375",
           "configs": {}}
    ctx = mp.get_context('spawn')
    for rate in [int(r) for r in a.rates.split(',')]:
        with tempfile.TemporaryDirectory() as tmp:
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):
                _scene(tmp, rate, a.frames)                     # write the scene files once
            with ctx.Pool(min(a.frames, budget)) as pool:
                one = pool.map(_one_frame, [(tmp, rate, a.frames, i, 'oracle') for i in range(a.frames)])
            P = max(1, min(10, budget))
            t0 = time.perf_counter()
            with ctx.Pool(P) as pool:
                many = pool.map(_one_frame, [(tmp, rate, a.frames, i % a.frames, 'oracle') for i in range(P)])
            wall = time.perf_counter() - t0
            with ctx.Pool(min(a.frames, budget)) as pool:
                cpp = pool.map(_one_frame, [(tmp, rate, a.frames, i, 'cpp') for i in range(a.frames)])
        secs = sorted(t for t, _ in one)
        med = secs[len(secs) // 2]
        nd = sum(n for _, n in one) / len(one)
        csecs = sorted(t for t, _ in cpp)
        res["configs"]["%d mm/hr" % rate] = {
            "streaks_per_frame": nd,
            "numpy_oracle_1core": {"seconds_per_frame": [round(t, 2) for t, _ in one], "median_s": med, "frames_per_s": 1.0 / med,
                                   "ms_per_drop": 1e3 * med / nd},
            "numpy_oracle_P_processes": {"P": P, "seconds_per_frame": [round(t, 2) for t, _ in many],
                                         "frames_per_s": sum(1.0 / t for t, _ in many), "wall_s_incl_startup": wall},
            "cpp_hostemu_1core": {"seconds_per_frame": [round(t, 2) for t, _ in cpp], "median_s": csecs[len(csecs) // 2],
                                  "frames_per_s": 1.0 / csecs[len(csecs) // 2]}}
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
