"""Host side of the driver alone (no GPU needed): per frame one 8-bit image + one 16-bit depth PNG decoded, the drop table
packed, two RGBA PNGs written from filtered scanlines -- the work Generator._run_batches gives its I/O threads around the
library call -- on T threads, for the writer's strategies.  Prints one JSON line.

    python scripts/host_io_bench.py [--threads 8] [--frames 96]"""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sub_rows(rgba):
    h, w = rgba.shape[:2]
    flat = rgba.reshape(h, 4 * w).astype(np.int16)
    rows = np.empty((h, 1 + 4 * w), np.uint8)
    rows[:, 0] = 1
    rows[:, 1:5] = flat[:, :4]
    rows[:, 5:] = ((flat[:, 4:] - flat[:, :-4]) & 255).astype(np.uint8)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--threads', type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument('--frames', type=int, default=96)
    ap.add_argument('--height', type=int, default=375)
    ap.add_argument('--width', type=int, default=1242)
    args = ap.parse_args()
    import __graft_entry__ as ge
    ge.build()
    from PIL import Image
    imgops = importlib.import_module('rain-rendering_amd.common.imgops')
    syn = importlib.import_module('rain-rendering_amd.synthetic')
    scenes = importlib.import_module('rain-rendering_amd.scenes')
    hb = scenes.hb
    H, W = args.height, args.width
    tmp = tempfile.mkdtemp(prefix='rainio_')
    nd = 16
    for i in range(nd):                                   # inputs the way a dataset holds them (PIL's adaptive filters)
        img = (syn.make_frame(i, H, W)[..., ::-1] * 255).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(tmp, 'i%02d.png' % i))
        d16 = (np.linspace(80, 2, H)[:, None] * np.ones((1, W)) * 256 + np.random.RandomState(i).uniform(0, 700, (H, W))).astype(np.uint16)
        Image.fromarray(d16).save(os.path.join(tmp, 'd%02d.png' % i))
    sc = scenes.Scene(os.path.join(tmp, 'scene'), H, W, 8192, n_frames=2, seed0=3000)
    table = list(sc.db.streaks_simulator.values())[0].table
    rgba = np.dstack([(syn.make_frame(3, H, W)[..., ::-1] * 255).astype(np.uint8), np.full((H, W), 255, np.uint8)])
    rows_img = sub_rows(rgba)
    lut = imgops.viridis_lut()                            # a mask: viridis(0) background, a few hundred streak-like blobs
    m = np.zeros((H, W))
    rng = np.random.RandomState(1)
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(400):
        cy, cx, ry, rx = rng.randint(H), rng.randint(W), rng.randint(8, 40), rng.randint(2, 8)
        m += np.exp(-(((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2)) * rng.uniform(0.1, 1)
    rows_mask = sub_rows(lut[np.clip((m / m.max() * 255).astype(int), 0, 255)])

    # the payloads as RR_OPT_PNG_DEFLATE delivers them: entropy-coded on the device (here: the host build of csrc/rr_deflate.h)
    import ctypes
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import helpers
    emu = helpers.hostemu()
    emu.emu_pngz.restype = ctypes.c_int64
    emu.emu_pngz.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    coded = []
    for r in (rows_img, rows_mask):
        r = np.ascontiguousarray(r, np.uint8).reshape(-1)
        d = np.zeros_like(r)
        assert emu.emu_pngz(r.ctypes.data, r.size, d.ctypes.data) > 0
        coded.append(d)

    def frame(i, strategy):
        bg = imgops.imread_bgr(os.path.join(tmp, 'i%02d.png' % (i % nd)))
        dep = imgops.imread_unchanged(os.path.join(tmp, 'd%02d.png' % (i % nd))).astype(np.float32) / 256.
        drops = hb.pack_frame(table, sc.db, W, H, i)
        a, m_ = (coded[0], coded[1]) if strategy is None else (rows_img, rows_mask)
        imgops.png_from_scanlines(os.path.join(tmp, 'o%d_a.png' % (i % 64)), a, W, H, strategy=strategy)
        imgops.png_from_scanlines(os.path.join(tmp, 'o%d_m.png' % (i % 64)), m_, W, H, strategy=strategy)
        return bg.shape[0] + dep.shape[0] + len(drops)

    res = {}
    for name, strategy in (('zlib_rle', 1), ('own_deflate', 3), ('payloads_coded_on_the_device', None)):
        with ThreadPoolExecutor(args.threads) as ex:
            list(ex.map(lambda i: frame(i, strategy), range(args.threads)))            # warm
            t = time.time()
            list(ex.map(lambda i: frame(i, strategy), range(args.frames)))
            dt = time.time() - t
        res[name] = {"frames_per_s": args.frames / dt, "cpu_ms_per_frame": 1e3 * dt * args.threads / args.frames}
    cpu = None
    try:
        cpu = [ln.split(':', 1)[1].strip() for ln in open('/proc/cpuinfo') if ln.startswith('model name')][0]
    except (OSError, IndexError):
        pass
    print(json.dumps({"what": "driver host work per frame without the GPU call: PNG decode (8-bit image + 16-bit depth, own inflate + "
                      "un-filtering), drop packing, two RGBA PNG files from filtered scanlines", "workload": "%dx%d, 8192 simulated streaks" % (W, H),
                      "threads": args.threads, "frames": args.frames, "cpu_model": cpu, "writers": res}))


if __name__ == '__main__':
    main()
