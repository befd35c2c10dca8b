"""The driver's HOST pipeline alone (no GPU needed): main.py on a synthetic on-disk dataset with the library's context
replaced by a stand-in that accepts every call and "delivers" pre-made PNG scanlines -- file walking, work list, PNG
decode, drop tables, the slot / batch logic of Generator._run_batches, deflate and file writes all run as in production;
only upload | kernels | download are absent.  What it measures: the ceiling the host side puts on the driver's frames/s
on this machine (codec + Python overhead), and with --tiny (frames of 64x48: negligible codec work) the ceiling of the
Python part alone.

    python scripts/driver_host_only.py [--frames 512] [--batch 128] [--tiny]
Prints one JSON line.  NOT a product path: the stand-in exists in this script only."""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))


class _Prepared:
    def __init__(self, frames, outs):
        self.frames, self.outs, self.n = frames, outs, len(frames)

    def set_drop_count(self, k, n):
        pass


class HostOnlyContext:
    """Accepts the calls Generator makes on RainHip; pipeline_wait fills the batch's scanline buffers from templates."""
    device = 0
    device_png = False                                # --device-png: the buffers hold zlib streams (RR_OPT_PNG_DEFLATE)

    def __init__(self, device=0):
        self.batches = {}
        self.rows = None

    def __getattr__(self, name):                      # set_camera, set_prepass_kernels, set_colormap, set_streak_db, ...
        if name.startswith('set_'):
            return lambda *a, **k: None
        raise AttributeError(name)

    def set_envmap_geometry(self, H, W, *tables):
        envmap = importlib.import_module('rain-rendering_amd.common.envmap')
        self.H, self.W = H, W
        return int(tables[0].shape[-1]) if hasattr(tables[0], 'shape') and tables[0].ndim >= 1 else W

    def host_rows(self, n, shape, dtype):
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        stride = max((nbytes + 15) // 16 * 16, 16)
        raw = np.zeros((n, stride), np.uint8)
        raw[...] = 1                                   # (touch every page now, like page-locked memory)
        return raw, [raw[k, :nbytes].view(dtype).reshape(shape) for k in range(n)]

    def host_free(self, raw):
        pass

    def pipeline_prepare(self, frames, outs):
        return _Prepared(frames, outs)

    def pipeline_submit_prepared(self, slot, prep, n=None):
        self.batches[slot] = (prep, prep.n if n is None else n)

    def pipeline_wait(self, slot):
        prep, n = self.batches.pop(slot, (None, 0))
        if prep is None:
            return True
        if self.rows is None:                         # templates: a sub-filtered image and a sub-filtered colour-mapped mask
            import host_io_bench as hib
            syn = importlib.import_module('rain-rendering_amd.synthetic')
            imgops = importlib.import_module('rain-rendering_amd.common.imgops')
            H, W = self.H, self.W
            rgba = np.dstack([(syn.make_frame(3, H, W)[..., ::-1] * 255).astype(np.uint8), np.full((H, W), 255, np.uint8)])
            m = np.zeros((H, W))
            rng = np.random.RandomState(1)
            yy, xx = np.mgrid[0:H, 0:W]
            for _ in range(max(4, H * W // 1200)):
                cy, cx, ry, rx = rng.randint(H), rng.randint(W), rng.randint(8, 40), rng.randint(2, 8)
                m += np.exp(-(((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2)) * rng.uniform(0.1, 1)
            lut = imgops.viridis_lut()
            self.rows = (hib.sub_rows(rgba).reshape(-1), hib.sub_rows(lut[np.clip((m / m.max() * 255).astype(int), 0, 255)]).reshape(-1))
            if HostOnlyContext.device_png:              # what RR_OPT_PNG_DEFLATE delivers: the zlib streams (host build of csrc/rr_deflate.h)
                import ctypes
                sys.path.insert(0, os.path.join(ROOT, 'tests'))
                import helpers
                emu = helpers.hostemu()
                emu.emu_pngz.restype = ctypes.c_int64
                emu.emu_pngz.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
                coded = []
                for r in self.rows:
                    r = np.ascontiguousarray(r, np.uint8)
                    d = np.zeros_like(r)
                    assert emu.emu_pngz(r.ctypes.data, r.size, d.ctypes.data) > 0
                    coded.append(d)
                self.rows = tuple(coded)
        if not getattr(prep, 'filled', False):         # (the "rendered" scanlines never change: written once per slot)
            for o in prep.outs:
                np.copyto(o['rainy_png'], self.rows[0])
                np.copyto(o['mask_png'], self.rows[1])
            prep.filled = True
        return True

    def close(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=512)
    ap.add_argument('--rate', type=int, default=100)
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--tiny', action='store_true', help='64x48 frames: the Python part alone')
    ap.add_argument('--distinct', type=int, default=32)
    ap.add_argument('--size', default=None, help='WxH of the dataset frames (default 1242x375)')
    ap.add_argument('--device-png', action='store_true', help='the stand-in delivers the PNG payloads entropy-coded, as the device does with RR_OPT_PNG_DEFLATE')
    ap.add_argument('--render-scale', type=int, default=1, help='render at 1/N of the frame size (what the Cityscapes plug-in does with N = 2)')
    args = ap.parse_args()
    os.environ['RAIN_BATCH'] = str(args.batch)
    import __graft_entry__ as ge
    ge.build()
    synthetic = importlib.import_module('rain-rendering_amd.synthetic')
    main_mod = importlib.import_module('rain-rendering_amd.main')
    generator_mod = importlib.import_module('rain-rendering_amd.common.generator')
    hb = importlib.import_module('rain-rendering_amd.hip_backend')
    hb.RainHip = HostOnlyContext                      # (this process only)
    HostOnlyContext.device_png = args.device_png
    H, W = (48, 64) if args.tiny else (375, 1242)
    if args.size:
        W, H = (int(v) for v in args.size.split('x'))
    if args.render_scale != 1:
        kitti = importlib.import_module('rain-rendering_amd.config.kitti')
        plain = kitti.settings
        kitti.settings = lambda: dict(plain(), render_scale=args.render_scale)
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, 'source')
        nd = min(args.distinct, args.frames)
        img_dir, dep_dir = synthetic.write_dataset(src, 'kitti', os.path.join('data_object', 'training'), nd, H, W, depth_m=None)
        for i in range(nd, args.frames):
            os.symlink(os.path.join(img_dir, '%06d.png' % (i % nd)), os.path.join(img_dir, '%06d.png' % i))
            os.symlink(os.path.join(dep_dir, '%06d.png' % (i % nd)), os.path.join(dep_dir, '%06d.png' % i))
        synthetic.write_streak_db(os.path.join(tmp, 'rainstreakdb'))
        frames = synthetic.simulate_particles(4, synthetic.DROPS_PER_RATE[args.rate], W // args.render_scale, H // args.render_scale)
        xml = os.path.join(tmp, 'particles', 'kitti', 'data_object', 'rain', '%dmm' % args.rate, 'sim_camera0.xml')
        synthetic.write_particles_xml(xml, frames)
        argv = ['--dataset', 'kitti', '-k', src, '-d', src, '-r', os.path.join(tmp, 'particles'), '-sd',
                os.path.join(tmp, 'rainstreakdb'), '-i', str(args.rate), '--output', os.path.join(tmp, 'out'), '--noverbose']
        t0 = time.time()
        gen = main_mod.main(argv)
        t1 = time.time()
        n = len(gen.stats)
        tm = gen.timing[0] if gen.timing else {}
        print(json.dumps({"what": "driver host pipeline without the GPU stages (stand-in context): decode, drop tables, batch logic, deflate, files",
                          "frames": n, "frame_size": [W, H], "seconds": t1 - t0, "frames_per_s": n / (t1 - t0),
                          "steady_frames_per_s": tm.get('steady_frames_per_s'), "cpu_quota": generator_mod._cpu_budget(),
                          "io_threads": gen._io_pool()._max_workers, "batch": args.batch}))


if __name__ == '__main__':
    main()
