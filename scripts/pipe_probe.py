"""Timeline probe of the host-pointer pipeline (rr_pipeline_submit / wait): run a few batches under
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python scripts/pipe_probe.py
and read, per batch, when its upload, its kernels and its download ran (scripts/pipe_timeline.py): do the three overlap?"""
import argparse
import importlib
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pipe-batch', type=int, default=32)
    ap.add_argument('--rounds', type=int, default=9)
    ap.add_argument('--copy-kernels', type=int, default=1)
    ap.add_argument('--drops', type=int, default=8192)
    ap.add_argument('--packed', type=int, default=1, help='frames of a slot back to back in one page-locked block per array; prepared descriptors')
    args = ap.parse_args()
    scenes = importlib.import_module('rain-rendering_amd.scenes')
    hb = scenes.hb
    fogmod = importlib.import_module('rain-rendering_amd.common.add_attenuation')
    envmod = importlib.import_module('rain-rendering_amd.common.envmap')
    imgops = importlib.import_module('rain-rendering_amd.common.imgops')
    H, W = 375, 1242
    PB = args.pipe_batch
    nd = min(PB, 16)                                   # distinct frames (the rest repeat them)
    sc = scenes.Scene(tempfile.mkdtemp(prefix='pipeprobe_'), H, W, args.drops, n_frames=nd, cam=scenes.KITTI)
    rh = hb.RainHip(0)
    rh.set_option(hb.RR_OPT_COPY_KERNELS, args.copy_kernels)
    rh.set_streak_db(sc.db.streaks_light)
    rh.set_camera(sc.cam)
    cs = sc.cam_settings
    consts = fogmod.FogRain(rain_intensity=100, focal=cs['focal_mm'] / 1000., f_number=cs['f_number'], angle=90, exposure=cs['exposure_ms'],
                            camera_gain=20).constants()
    rh.set_prepass_kernels(imgops.gaussian_kernel(25, 25), imgops.gaussian_kernel(15, 0))
    rh.set_envmap_geometry(H, W, *envmod.EnvironmentMapGenerator(cs['focal_mm'] / 1000., W, H).device_tables(H, W))
    depth = (np.linspace(80, 2, H, dtype=np.float32)[:, None] * np.ones((1, W), np.float32))
    host = [(sc.frame_inputs(i)[0], sc.product_drops(i)) for i in range(nd)]
    rh.set_solid_angles(sc.omega)
    cap = (max(len(h_[1]) for h_ in host) + 3) // 4 * 4
    slots = []
    nslot = hb.RR_PIPE_SLOTS
    for s_ in range(nslot):
        arrs = [rh.host_rows(PB, shp, dt)[1] for shp, dt in (((H, W, 3), np.uint8), ((H, W), np.float32), ((cap,), hb.DROP_DTYPE),
                                                           ((H, W, 3), np.uint8), ((H, W), np.int32))]
        bg8s, deps, drs, ims, mks = arrs
        frs, outs, nds = [], [], []
        for k in range(PB):
            bg, dr_ = host[(s_ * PB + k) % nd]
            bg8s[k][...] = (bg * 255).astype(np.uint8)
            deps[k][...] = depth
            drs[k][:len(dr_)] = dr_
            nds.append(len(dr_))
            frs.append(dict(bg_u8=bg8s[k], depth=deps[k], fog=consts, omega=None, drops=drs[k]))
            outs.append(dict(image_u8=ims[k], mask_i32=mks[k]))
        prep = rh.pipeline_prepare(frs, outs)
        for k, n_ in enumerate(nds):
            prep.set_drop_count(k, n_)
        slots.append(prep)

    def pipe(rounds):
        t_sub, t_wait = 0.0, 0.0
        for r in range(rounds + nslot):
            s_ = r % nslot
            if r >= nslot:
                a = time.perf_counter()
                while not rh.pipeline_wait(s_):
                    rh.pipeline_submit_prepared(s_, slots[s_])
                t_wait += time.perf_counter() - a
            if r < rounds:
                a = time.perf_counter()
                rh.pipeline_submit_prepared(s_, slots[s_])
                t_sub += time.perf_counter() - a
        return t_sub, t_wait
    pipe(nslot)
    t0 = time.perf_counter()
    t_sub, t_wait = pipe(args.rounds)
    el = time.perf_counter() - t0
    print("PIPE frames/s %.0f  (%d x %d frames in %.1f ms; host time inside submit %.1f ms, inside wait %.1f ms)"
          % (args.rounds * PB / el, args.rounds, PB, 1e3 * el, 1e3 * t_sub, 1e3 * t_wait))
    rh.close()


if __name__ == '__main__':
    main()
