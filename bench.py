"""bench.py -- rainy frames/sec of the MI355X hot path on synthetic inputs shaped like BASELINE.json's configs.

  python bench.py --gpus N --steps K --warmup W            (N=1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (rr_render_frames_device: FOV spans + sums, per-drop plan, colour, tile synthesis,
defocus blur, ordered compositing, finalise) over one batch of --batch synthetic frames whose inputs are already
resident in HBM.  Workload (default) = BASELINE.json configs[2]: 1242x375, "100 mm/hr" = 8192 streaks per frame
(synthetic count); --workload cityscapes50 = configs[3]: 2048x1024, 4096 streaks, 5 ms exposure.
Frames shard across ranks; the only collective on the data path is one RCCL broadcast of the packed streak
database from rank 0 at start-up.  Rank 0 prints ONE JSON line.

  default            weak scaling: every rank renders its own --batch frames per step
  --total-frames T   strong scaling: ONE T-frame sequence, frames idx[rank::world] per rank (sharding.shard), a step
                     renders the rank's share in chunks of --batch; value = T * steps / time

Extra keys (N=1 only, outside the timed region, never `value`): `host_inclusive` (pinned, three-stream
rr_pipeline_submit/wait incl. PCIe both ways and the fog / environment-map pre-pass), `prepass`, `variants`
(no tile sharing, angular noise, smaller library calls), `chain` (whole-step HBM fraction), `cpu_baseline` (+ its
multi-process and C++ legs).  roofline.traffic is measured by two rocprofv3 --pmc passes of this very workload that
bench.py spawns itself (skip with --no-traffic).
"""
import argparse
import contextlib
import csv
import ctypes
import glob
import importlib
import json
import os
import re
import shutil
import signal
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: frame size, rate -> synthetic drop count, camera preset, render_scale, BASELINE.json config
    'kitti100': dict(H=375, W=1242, rate=100, cam='KITTI', rs=1, cfg='configs[2]', metric="rainy frames/sec @ 1242x375, 100 mm/hr"),
    'kitti25': dict(H=375, W=1242, rate=25, cam='KITTI', rs=1, cfg='configs[1]', metric="rainy frames/sec @ 1242x375, 25 mm/hr"),
    'cityscapes50': dict(H=1024, W=2048, rate=50, cam='CITYSCAPES', rs=1, cfg='configs[3]', metric="rainy frames/sec @ 2048x1024, 50 mm/hr"),
    # the reference's default for Cityscapes: render_scale = depth_scale = 2 (config/cityscapes.py:41-42) -> 1024x512 frames
    'cityscapes50_rs2': dict(H=512, W=1024, rate=50, cam='CITYSCAPES', rs=2, cfg='configs[3] at the plug-in\'s default render_scale 2',
                             metric="rainy frames/sec @ 1024x512 (2048x1024 / render_scale 2), 50 mm/hr"),
}
# BASELINE.json configs[4]: nuScenes 1600x900, {1, 5, 25, 100, 200} mm/hr, IN-KERNEL particle simulation (no XML): the drop tables
# are generated on the device inside the timed step (rr_generate_drops_device), at the physical (Poisson) counts of
# tools/particles.py's model; nuscenes200x = the heaviest rain at SURVEY 8d's fixed 16384 particles per frame
for _r in (1, 5, 25, 100, 200):
    WORKLOADS['nuscenes%d' % _r] = dict(H=900, W=1600, rate=_r, cam='NUSCENES', rs=1, cfg='configs[4]', sim=True, count=None,
                                        metric="rainy frames/sec @ 1600x900, %d mm/hr, in-kernel particles" % _r)
WORKLOADS['nuscenes200x'] = dict(H=900, W=1600, rate=200, cam='NUSCENES', rs=1, cfg='configs[4]', sim=True, count=16384,
                                 metric="rainy frames/sec @ 1600x900, 200 mm/hr (16384 particles/frame), in-kernel particles")


F64_VECTOR_PEAK_TFLOPS = 78.6    # MI355X float64 vector peak, FMA = 2 flops: half of MI355X_MICROARCH.md's 157.3 TFLOP/s float32
                                 # vector figure (a wave64 float64 instruction holds its SIMD-32 for 4 cycles, a float32 one for 2)


def blur_flop_model(drops, status, cam, H, W):
    """Float64 operations of the defocus blur of one frame, per kernel, from the drop records (the arithmetic of
    rr_device.h plan_drop restated with numpy: tile size, circle of confusion, radii).  A filter output of radius r costs
    1 + 3 r operations (centre product; per tap one add of the two symmetric samples, one product, one add into the sum --
    no FMA contraction on the alpha path: the additions and products are separate instructions by contract).  Which kernel
    takes a drop: blur_is_small (rr_device.h)."""
    d = drops[status == 0]
    big = d['type'] == 0
    d0, d1 = np.floor(d['iw1']), np.floor(d['iw2'])
    minx = np.maximum(np.minimum(d['x0'], d['x1']), 0)
    miny = np.maximum(np.minimum(d['y0'], d['y1']), 0)
    maxx = np.minimum(np.maximum(d['x0'] + d0, d['x1'] + d1), W)
    maxy = np.minimum(np.maximum(d['y0'], d['y1']), H)
    tw = np.where(big, np.maximum((maxx - minx).astype(np.int64), 1), np.maximum(np.abs(d['x1'] - d['x0']), d['max_width'] + 2))
    th = np.where(big, np.maximum(maxy - miny, 1), np.maximum(np.abs(d['y1'] - d['y0']), 2))
    o = np.abs(d['wps'][:, 2])
    cc = np.abs(((o - cam.focus_plane) * cam.focal_sq) / (o * (cam.focus_plane - cam.focal_m) * cam.f_number) / cam.sensor_px)
    r1 = np.where(cc > 1e-15, (4.0 * cc + 0.5).astype(np.int64), 0)
    r2 = np.where(cc / 2 > 1e-15, (2.0 * cc + 0.5).astype(np.int64), 0)
    ew, eh = tw + 2 * r2, th + 2 * r1
    flops = tw * eh * (1 + 3 * r1) + np.where(r2 > 0, ew * eh * (1 + 3 * r2), 0)
    php = (eh + 3) & ~3
    ypitch = (((ew + 3) & ~3) + 2 * r2) | 1
    small = (r1 > 0) & (r1 <= 31) & (tw * (php + 2 * r1) <= 512) & (ypitch * php <= 768)
    rest = (r1 > 0) & ~small
    return {"k_blur_small": float(flops[small].sum()), "k_blur_fused": float(flops[rest & (r1 <= 48)].sum()),
            "k_blur_rows": float(flops[rest & (r1 > 48)].sum()), "blurred_drops": int((r1 > 0).sum()), "output_px": float((ew * eh)[r1 > 0].sum())}


def algorithmic_bytes(H, W, He, We, N):
    """bytes(frame) = 27*H*W + 16*He*We + 64*N."""
    return 27 * H * W + 16 * He * We + 64 * N


class DeviceBatch:
    """n frames resident in HBM as torch tensors + the ctypes descriptors rr_render_frames_device takes."""

    def __init__(self, torch, hb, sc, dev, frame_ids, drop_ids, noise_std=0.0, sims=None, in_dtype='f32'):
        """sims: SIM_FRAME_DTYPE records, one per frame: the drop tables are generated on the device (generate()), `drops`
        and the drop counts live in HBM only.  in_dtype: element type of the image / map inputs in HBM ('f32': float32 image,
        float32 xyY map and solid angles, rr_frame_in.in_types -- what the colour channels need; 'f64': the reference's arrays)."""
        self.n = len(frame_ids)
        f32 = in_dtype == 'f32'
        self.in_types = (hb.RR_IN_BG_F32 | hb.RR_IN_ENV_F32) if f32 else 0
        self.sims = sims
        if sims is not None:
            self.cap = max(int(sims['n_particles'].max()), 1)
            self.t_drops = torch.empty((self.n, self.cap * hb.DROP_DTYPE.itemsize), dtype=torch.uint8, device=dev)
            self.t_counts = torch.zeros((self.n,), dtype=torch.int32, device=dev)
        self.keep = []
        self.status_tensors = []
        self.fin = (hb.rr_frame_in * max(self.n, 1))()
        self.fout = (hb.rr_frame_out * max(self.n, 1))()
        self.host = []
        H, W, He, We = sc.H, sc.W, sc.He, sc.We
        omega_t = torch.from_numpy(np.ascontiguousarray(sc.omega, np.float32 if f32 else np.float64)).to(dev)
        self.keep.append(omega_t)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max(1, min(8, os.cpu_count() or 1))) as pool:       # (seeded per frame: the order of evaluation is free)
            inputs = list(pool.map(sc.frame_inputs, frame_ids))
        for k, (fi, di) in enumerate(zip(frame_ids, drop_ids)):
            bg, env = inputs[k]
            inputs[k] = None
            if sims is None:
                drops = sc.product_drops(di, noise_std=noise_std, noise_scale=1.0 if noise_std else 0.0)
                t_dr = torch.from_numpy(drops.view(np.uint8).reshape(-1)).to(dev)
            else:
                drops = np.zeros(self.cap, hb.DROP_DTYPE)           # (length = capacity; the records are made on the device)
                t_dr = self.t_drops[k]
            self.host.append((bg, env if k < 8 else None, drops))      # (host copies of the maps: the CPU legs use frame 0)
            t_bg = torch.from_numpy(bg.astype(np.float32) if f32 else bg).to(dev)
            t_env = torch.from_numpy(env.astype(np.float32) if f32 else env).to(dev)
            o_rgb = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
            o_m = torch.empty((H, W), dtype=torch.float64, device=dev)
            o_mi = torch.empty((H, W), dtype=torch.int32, device=dev)
            o_st = torch.empty((max(len(drops), 1),), dtype=torch.int32, device=dev)
            self.keep += [t_bg, t_env, t_dr, o_rgb, o_m, o_mi, o_st]
            self.status_tensors.append(o_st)
            fi_, fo_ = self.fin[k], self.fout[k]
            fi_.H, fi_.W, fi_.He, fi_.We = H, W, He, We
            fi_.bg = fi_.rainy_bg = t_bg.data_ptr()
            fi_.env_xyY = t_env.data_ptr()
            fi_.omega = omega_t.data_ptr()
            fi_.in_types = self.in_types
            fi_.drops = t_dr.data_ptr()
            fi_.n_drops = len(drops)
            if sims is not None:
                fi_.n_drops_dev = self.t_counts[k:k + 1].data_ptr()
            fi_.strategy = 0
            fi_.opacity_attenuation = 1.0
            fo_.rainy_rgb = o_rgb.data_ptr()
            fo_.rainy_bg_out = None
            fo_.mask_f64 = o_m.data_ptr()
            fo_.mask_i32 = o_mi.data_ptr()
            fo_.drop_status = o_st.data_ptr()
        self.mean_drops = float(np.mean([len(f[2]) for f in self.host])) if self.host else 0.0

    def generate(self, rh, H, W, stream, a=0, b=None):
        """Drop tables of frames [a, b) on the device (rr_generate_drops_device), on the launch stream."""
        b = self.n if b is None else b
        rh.generate_drops_device(self.sims[a:b], H, W, self.t_drops[a].data_ptr(), self.cap, self.t_counts[a:b].data_ptr(), stream)

    def device_counts(self):
        return self.t_counts.cpu().numpy()

    def chunk(self, hb, a, b):
        """Descriptor arrays of frames [a, b) (they point into the same tensors)."""
        n = b - a
        fin, fout = (hb.rr_frame_in * n)(), (hb.rr_frame_out * n)()
        for k in range(n):
            ctypes.memmove(ctypes.byref(fin[k]), ctypes.byref(self.fin[a + k]), ctypes.sizeof(hb.rr_frame_in))
            ctypes.memmove(ctypes.byref(fout[k]), ctypes.byref(self.fout[a + k]), ctypes.sizeof(hb.rr_frame_out))
        return fin, fout, n


def timed(torch, dist, world, dev, fn, steps):
    """K calls of fn bracketed by barrier + synchronize on both sides; MAX over ranks."""
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    return el


def measure_traffic(args, dom_kernels, scene_dir=None):
    """HBM bytes per launch of the dominant kernel: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they do not
    fit one pass) of a short run of this same workload, corrected as MI355X_MICROARCH.md's HBM section prescribes:
    KB * 1024, FETCH_SIZE doubled on gfx950.  None when rocprofv3 is missing or anything goes wrong."""
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    vals = {}
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = tempfile.mkdtemp(prefix='rainpmc_', dir='/tmp')
            cmd = [exe, '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', out, '--', sys.executable,
                   os.path.abspath(__file__), '--inner', '--steps', '2', '--warmup', '1', '--batch', str(args.batch),
                   '--workload', args.workload, '--input-dtype', args.input_dtype] + sum((['--opt', o] for o in args.opt), []) + ['--opt', '21=0']      # (counters per kernel: every kernel alone on the device)
            if scene_dir:
                cmd += ['--scene-dir', scene_dir]                   # the simulation this process already wrote
            env = dict(os.environ, TMPDIR='/tmp')
            # own session + a hard limit: a profiler that aborts can sit on its child for ever; the whole group is killed
            proc = subprocess.Popen(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=300)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)
                proc.wait()
                return None, "pmc pass timed out"
            if rc != 0:
                return None, "pmc pass exited with %d" % rc
            tot, cnt = 0.0, 0
            for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
                for row in csv.DictReader(open(f)):
                    m = re.search(r'(k_[a-z_0-9]+)', row.get('Kernel_Name', ''))
                    if m and (m.group(1) in dom_kernels or scope_of_name(row.get('Kernel_Name', '')) in dom_kernels) and row['Counter_Name'] == counter:
                        tot += float(row['Counter_Value'])
                        cnt += 1
            shutil.rmtree(out, ignore_errors=True)
            if not cnt:
                return None, "kernel not found in the counter file"
            vals[counter] = tot / cnt * len(dom_kernels)            # a timing scope may hold several kernels
        return ((2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0,
                "(2*FETCH_SIZE + WRITE_SIZE) KB * 1024 per launch, two rocprofv3 --pmc passes of this workload spawned by bench.py")
    except Exception as e:                                          # noqa: BLE001 -- reported, never fatal
        return None, "pmc pass failed: %r" % (e,)


# defaults of the library's rr_set_option switches (what a sweep puts back)
OPTION_DEFAULTS = {1: 1, 9: 1, 10: 2, 12: 1, 13: 1, 16: 1, 17: 1, 18: 1, 19: 1, 20: 1, 21: 1, 22: 1, 23: 2}

# kernel symbol -> the name of the timing scope (rr_profile_read) it is launched under
SCOPE_OF = {'k_blur_fused_dma': 'k_blur_fused', 'k_composite32': 'k_composite', 'k_fov_sums32': 'k_fov_sums', 'k_fov_dda': 'k_fov_spans',
            'k_finalize16': 'k_finalize', 'k_bin_rows': 'k_bin', 'k_env_consts': 'k_env_prefix',
            'k_fov_poly_general': 'k_fov_poly', 'k_png_image': 'k_png_rows', 'k_png_mask': 'k_png_rows', 'k_pngz_blocks': 'k_pngz',
            'k_pngz_pack': 'k_pngz', 'k_rows_scatter': 'k_lists', 'k_rows_shares': 'k_lists', 'k_plan_big': 'k_plan'}


def scope_of(kernel):
    return SCOPE_OF.get(kernel, kernel)


def scope_of_name(full_name):
    """Timing scope of a kernel from its (demangled) name as rocprofv3 prints it; '' when it is none of the library's."""
    m = re.search(r'(k_[a-z_0-9]+)(<[^>]*>)?', full_name or '')
    if not m:
        return ''
    if m.group(1) == 'k_blur' and m.group(2) in ('<0>', '<1>'):              # the two-pass blur: one kernel template, a scope per pass
        return 'k_blur_rows' if m.group(2) == '<0>' else 'k_blur_cols'
    return scope_of(m.group(1))


def measure_valu(args, scene_dir=None):
    """VALU issue utilisation per kernel from one rocprofv3 --pmc pass of this workload.  SQ_INSTS_VALU counts wave-level
    VALU instructions over the whole chip; a wave64 instruction occupies its SIMD-32 for 2 cycles (float / integer) or 4
    (float64: half rate), so the fraction of the chip's VALU issue capacity in use while the kernel runs lies between
        2 * SQ_INSTS_VALU / (cycles * 1024 SIMDs)   and   4 * SQ_INSTS_VALU / (cycles * 1024 SIMDs),
    cycles = GRBM_GUI_ACTIVE / 8 (the counter adds up the 8 XCDs; checked against the kernel's duration).  The path's
    arithmetic is float64 almost everywhere: the upper figure is the closer one.  `waiting` = SQ_WAIT_ANY / SQ_WAVE_CYCLES:
    the share of wave time spent parked on s_waitcnt / barriers.  Returns ({kernel: {...}}, how)."""
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    try:
        out = tempfile.mkdtemp(prefix='rainpmc_', dir='/tmp')
        cmd = [exe, '--kernel-trace', '--pmc', 'SQ_INSTS_VALU', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_BUSY_CYCLES', 'GRBM_GUI_ACTIVE',
               '--output-format', 'csv', '-d', out, '--', sys.executable, os.path.abspath(__file__), '--inner', '--steps', '2', '--warmup', '1',
               '--batch', str(args.batch), '--workload', args.workload, '--input-dtype', args.input_dtype] + sum((['--opt', o] for o in args.opt), []) + ['--opt', '21=0']      # (counters per kernel: every kernel alone on the device)
        if scene_dir:
            cmd += ['--scene-dir', scene_dir]
        proc = subprocess.Popen(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                start_new_session=True)
        try:
            rc = proc.wait(timeout=300)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            proc.wait()
            return None, "pmc pass timed out"
        if rc != 0:
            return None, "pmc pass exited with %d" % rc
        agg, cnt = {}, {}
        for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
            for row in csv.DictReader(open(f)):
                m = re.search(r'(k_[a-z_0-9]+)', row.get('Kernel_Name', ''))
                if not m:
                    continue
                key = (m.group(1), row['Counter_Name'])
                agg[key] = agg.get(key, 0.0) + float(row['Counter_Value'])
                cnt[key] = cnt.get(key, 0) + 1
        shutil.rmtree(out, ignore_errors=True)
        res = {}
        for (k, c) in agg:
            res.setdefault(k, {})[c] = agg[(k, c)] / cnt[(k, c)]
        outp = {}
        for k, v in res.items():
            if v.get('GRBM_GUI_ACTIVE') and v.get('SQ_INSTS_VALU') is not None:
                cyc = v['GRBM_GUI_ACTIVE'] / 8.0
                outp[k] = {"valu_util": 4.0 * v['SQ_INSTS_VALU'] / (cyc * 1024.0), "valu_util_if_all_f32": 2.0 * v['SQ_INSTS_VALU'] / (cyc * 1024.0),
                           "valu_insts": v['SQ_INSTS_VALU'], "cycles": cyc,
                           "waiting": (v.get('SQ_WAIT_ANY', 0.0) / v['SQ_WAVE_CYCLES']) if v.get('SQ_WAVE_CYCLES') else None}
        if not outp:
            return None, "no counters in the pass' output"
        return outp, ("valu_util = 4 * SQ_INSTS_VALU / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs): the share of the chip's VALU issue capacity in use if "
                      "every instruction were float64 (4 cycles per wave64 instruction; 2 for float / integer: half the figure is the lower "
                      "bound); waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES; one rocprofv3 --pmc pass of this workload spawned by bench.py")
    except Exception as e:                                          # noqa: BLE001 -- reported, never fatal
        return None, "pmc pass failed: %r" % (e,)


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_baseline(sc, host, W, H, sample_drops, procs):
    """The CPU legs (rank 0, N=1 only).  `port`: the numpy oracle in its op-for-op mode (per-drop masked reduction
    over the whole environment map, like the reference) on ONE core, on a bounded sample (the first `sample_drops`
    streaks of frame 0), extrapolated linearly in the drop count.  `processes`: P copies of that sample on P cores
    (reference main_threaded.py:176 caps its pool at 10).  `cpp`: the g++ -O2 build of the kernel arithmetic
    (tests/hostemu), whole frame, one core -- the stronger CPU baseline."""
    from oracle import render as orc                 # the checker, timed: the only place bench.py touches oracle/
    bg, env, drops = host[0]
    textures, ratio = orc.load_streak_database(sc.tex_dir, sc.norm)
    sim0 = list(orc.load_streaks_from_xml(sc.xml, sc.render_scale, [W, H]).values())[0]
    streaks = list(orc.streak_filter(sim0.streaks, W, H).values())
    ns = min(sample_drops, len(streaks))
    c0 = time.perf_counter()
    orc.render_frame(bg, bg, env, sc.omega, streaks, textures, ratio, sc.ocam, frame_seed=0, faithful=True, max_drops=ns)
    c1 = time.perf_counter()
    per_drop = (c1 - c0) / ns
    out = {"value": 1.0 / (per_drop * len(streaks)), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "numpy oracle (op-for-op), first %d of %d streaks of frame 0: %.1f s, %.2f ms/drop, extrapolated linearly "
                     "in the drop count; whole-frame medians: profiles/ (scripts/cpu_baseline_full.py)"
                     % (ns, len(streaks), c1 - c0, 1e3 * per_drop),
           "nproc": os.cpu_count(), "cpu_model": _cpu_model()}
    try:                                                            # P processes, same sample each
        import multiprocessing as mp
        P = max(1, min(10, procs))
        ctx = mp.get_context('fork')
        q = ctx.Queue()

        def work(k):
            import copy
            t0 = time.perf_counter()
            orc.render_frame(bg, bg, env, sc.omega, copy.deepcopy(streaks[:ns]), textures, ratio, sc.ocam, frame_seed=k, faithful=True)
            q.put(time.perf_counter() - t0)
        p0 = time.perf_counter()
        ps = [ctx.Process(target=work, args=(k,)) for k in range(P)]
        for p in ps:
            p.start()
        for p in ps:
            p.join()
        wall = time.perf_counter() - p0
        out["processes"] = {"value": P / (wall / ns * len(streaks)), "unit": "frames/s", "cores": P,
                            "sample": "%d processes (min(10, nproc), main_threaded.py:176) x the same %d-streak sample: %.1f s wall" % (P, ns, wall)}
    except Exception as e:                                          # noqa: BLE001
        out["processes"] = {"error": repr(e)}
    try:                                                            # C++ build of the kernel arithmetic
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        import helpers as th
        th.hostemu()
        e0 = time.perf_counter()
        th.emu_render(sc, bg, bg, env, drops)
        e1 = time.perf_counter()
        out["cpp"] = {"value": 1.0 / (e1 - e0), "unit": "frames/s", "cores": 1, "kind": "port",
                      "sample": "tests/hostemu (rr_device.h compiled with g++ -O2 -ffp-contract=off), whole frame 0 (%d streaks): %.2f s"
                                % (len(drops), e1 - e0)}
    except Exception as e:                                          # noqa: BLE001
        out["cpp"] = {"error": repr(e)}
    return out


def driver_end_to_end(frames, batch):
    """The drop-in driver (main.py) on an on-disk synthetic dataset of the headline shape, PNG in -> PNG out, in its own
    process after the timed region (scripts/driver_e2e.py): reported beside the headline, never `value`.  Host-bound (PNG
    codec on the box's CPU quota)."""
    import subprocess
    cmd = [sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'scripts', 'driver_e2e.py'), '--frames', str(frames),
           '--batch', str(batch)]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=150, env=dict(os.environ, RAINHIP_OPTIONS=''))
        d = json.loads(r.stdout.decode().strip().splitlines()[-1])
        tm = (d.get('timing') or [{}])[0]
        return {"what": d.get('what'), "frames": d.get('frames'), "frames_per_s_steady": tm.get('steady_frames_per_s'),
                "frames_per_s_including_setup": d.get('frames_per_s'), "host_route": tm.get('route', 'general'),
                "cpu_quota": d.get('cpu_quota'), "frames_per_batch": batch, "workload": d.get('workload')}
    except Exception as e:                                # (a reported extra must not take the headline down with it)
        return {"error": "%s: %s" % (type(e).__name__, e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=512, help='frames per library call (and, in weak scaling, per step and GPU); raw tiles '
                    'that are bit-identical across the frames of a call are rendered once: 256 -> 512 frames per call is +9 %% frames/s')
    ap.add_argument('--workload', choices=sorted(WORKLOADS), default='kitti100')
    ap.add_argument('--total-frames', type=int, default=0, help='strong scaling: one sequence of this many frames sharded over the ranks')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-prepass', action='store_true', help='skip the pre-pass / host-inclusive extras')
    ap.add_argument('--no-variants', action='store_true')
    ap.add_argument('--more-variants', action='store_true', help='further host-inclusive variants (slot sizes, buffer layouts)')
    ap.add_argument('--no-traffic', action='store_true')
    ap.add_argument('--copy-sweep', default='', help='comma-separated workgroup counts of the batched copy kernels to time the host-inclusive leg with')
    ap.add_argument('--no-driver', action='store_true', help='skip the main.py driver end-to-end leg (PNG in -> PNG out, own process)')
    ap.add_argument('--cpu-sample-drops', type=int, default=2048)
    ap.add_argument('--pipe-batch', type=int, default=128, help='frames per slot of the host-inclusive pipeline (the driver\'s default batch)')
    ap.add_argument('--input-dtype', choices=('f32', 'f64'), default='f32', help='element type of the image / map inputs resident in HBM '
                    '(rr_frame_in.in_types); f64 = the reference\'s float64 arrays (timed as a variant)')
    ap.add_argument('--opt', action='append', default=[], help='rr_set_option as ID=VALUE (tuning switches that never change results)')
    ap.add_argument('--sweep', action='append', default=[], help='A/B: after the headline, time the loop again under these '
                    'rr_set_option sets ("6=3,3=512"); one JSON line each on stderr; implies the lean run')
    ap.add_argument('--scene-dir', default=None, help='(used by the PMC passes) directory of a simulation to reuse')
    ap.add_argument('--no-serial', action='store_true', help='skip the extra pass with the colour branch on the main stream (per-kernel durations without overlap)')
    ap.add_argument('--phases', action='store_true', help='(phase-clock build of the library) print the per-phase wave cycles')
    ap.add_argument('--inner', action='store_true', help='(used by the PMC passes) timed loop only, no extras, no JSON')
    args = ap.parse_args()
    if args.inner or args.sweep:
        args.no_cpu_baseline = args.no_prepass = args.no_variants = args.no_traffic = args.no_driver = True

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # test hooks (one-GPU smoke test of the N>1 path): RAIN_BENCH_DEVICE pins every rank to one device,
    # RAIN_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU)
    if 'RAIN_BENCH_DEVICE' in os.environ:
        local_rank = int(os.environ['RAIN_BENCH_DEVICE'])
    backend = os.environ.get('RAIN_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        dist.barrier()
    scenes = importlib.import_module('rain-rendering_amd.scenes')
    sharding = importlib.import_module('rain-rendering_amd.sharding')
    hb, synthetic = scenes.hb, scenes.synthetic

    wl = WORKLOADS[args.workload]
    H, W, B = wl['H'], wl['W'], args.batch
    is_sim = bool(wl.get('sim'))
    N = 16 if is_sim else synthetic.DROPS_PER_RATE[wl['rate']]      # (in-kernel particles: the scene's XML is a stub, never rendered)
    cam = getattr(scenes, wl['cam'])
    strong = args.total_frames > 0
    # frames of this rank: weak = its own B frames (seeded by rank); strong = its share of ONE sequence
    if strong:
        my_frames = sharding.shard(list(range(args.total_frames)), rank, world)
        n_sim, seed0 = args.total_frames, 3000
    else:
        my_frames = list(range(B))
        n_sim, seed0 = B, 3000 + 1000 * rank
    tmp = args.scene_dir or tempfile.mkdtemp(prefix='rainbench_r%d_' % rank)
    with contextlib.redirect_stdout(sys.stderr):       # loaders print like the reference's do; stdout carries the JSON line only
        sc = scenes.Scene(tmp, H, W, N, n_frames=1 if is_sim else n_sim, cam=cam, seed0=seed0, render_scale=wl['rs'])
    He, We = sc.He, sc.We
    sims = None
    if is_sim:
        particles = importlib.import_module('rain-rendering_amd.tools.particles')
        sdb = importlib.import_module('rain-rendering_amd.common.db')
        opt = dict(sdb.settings('nuscenes'))
        opt.pop('sequences', None)
        # frame numbers of this rank's share of the sequence: particle (seed, frame, index) is the same whoever makes it
        fnum = my_frames if strong else [f + 100000 * rank for f in my_frames]
        sims, dgrid, cdf = particles.sim_frames(opt, wl['rate'], len(my_frames), render_scale=wl['rs'], seed=2024, count=wl['count'],
                                                frame_ids=fnum, draw_seeds=[f % (2 ** 32) for f in fnum])

    rh = hb.RainHip(local_rank)
    user_opts = {}
    for kv in args.opt:
        k, v = kv.split('=')
        rh.set_option(int(k), int(v))
        user_opts[int(k)] = int(v)
    # --- the one collective: RCCL broadcast of the packed streak DB over xGMI -----------------
    texels, hs, ws, offs = hb.pack_streak_db(sc.db.streaks_light)
    t_tex = torch.from_numpy(texels).to(dev)
    if world > 1:
        dist.broadcast(t_tex, src=0)
        torch.cuda.synchronize()
    rh.set_streak_db_device(t_tex.data_ptr(), t_tex.numel(), hs, ws, offs)
    rh.set_camera(sc.cam)
    if is_sim:
        rh.set_particle_tables(dgrid, cdf)

    # --- inputs resident in HBM ----------------------------------------------------------------
    fids = [f if strong else f + 100 * rank for f in my_frames]
    batch = DeviceBatch(torch, hb, sc, dev, fids, my_frames, sims=sims, in_dtype=args.input_dtype)
    stream = torch.cuda.current_stream().cuda_stream
    chunks = [batch.chunk(hb, a, min(a + B, batch.n)) for a in range(0, batch.n, B)]
    bounds = [(a, min(a + B, batch.n)) for a in range(0, batch.n, B)]

    def render(chs=None):
        for ci, (fin, fout, n) in enumerate(chunks if chs is None else chs):
            if is_sim:                 # in-kernel particle simulation: the drop tables of the call's frames are (re)made first
                batch.generate(rh, H, W, stream, *bounds[ci])
            rh.render_frames_device(fin, fout, n, stream)

    def warm(fn, reps):
        for _ in range(max(reps, 1)):          # also sizes the tile arena: re-enqueue until it fits
            fn()
            while not rh.synchronize():
                fn()
        torch.cuda.synchronize()

    warm(render, args.warmup)
    rh.profile_reset()
    rh.profile(True)               # HIP events around every launch, on the launch stream
    elapsed = timed(torch, dist, world, dev, render, args.steps)
    rh.profile(False)
    assert rh.synchronize(), "tile arena regrew inside the timed region"
    stats = rh.profile_read()
    if args.phases and hasattr(rh.lib, 'rr_debug_phases'):         # phase-clock build of the library (scripts/phase_timing.sh)
        buf = (ctypes.c_ulonglong * 64)()
        rh.lib.rr_debug_phases.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        rh._check(rh.lib.rr_debug_phases(rh.h, buf, 1), 'rr_debug_phases')
        names = {0: ('k_tile', ['stage plan+texture', 'row intervals', 'samples', 'horizontal folds', 'wait for waves', 'vertical folds+store', '-', 'barrier at item start']),
                 1: ('k_tile_big', ['search+barriers', 'plan fields', 'pixel']),
                 2: ('k_blur_small', ['plan+weights+raw->LDS', 'row pass', 'column pass+store']),
                 3: ('k_blur_fused[_dma]', ['issue loads | dma: clear Y + wait for the loads + halo', 'barrier A (loads land)', 'row pass', 'barrier B', 'column pass+store', 'barrier C', 'dma: next loads issued']),
                 5: ('k_tile_rows', ['pull + plan', 'column table', 'group set-up (clear, row terms, intervals)', 'row walks', 'vertical folds+store', 'texture switch (wait + stage)', 'COUNT walk iterations', 'Big tiles (bicubic warp, a lane per pixel)']),
                 4: ('k_composite32', ['background / depth loads', 'list piece: clist + bbox loads, ballots, list', 'barriers of the list piece', 'record batch arrives', 'entry loop',
                                       'barrier at the piece end', 'stores + tile reduction'])}
        calls = args.steps + args.warmup
        sys.stderr.write("PHASES (shader cycles summed over waves, per call; share of the kernel's wave time)\n")
        for kid, (kn, ph) in names.items():
            row = [int(buf[kid * 8 + q]) for q in range(8)]
            tot = sum(row) or 1
            sys.stderr.write("  %-14s %s\n" % (kn, ', '.join('%s %.1f%% (%.3g)' % (ph[q] if q < len(ph) else '?', 100.0 * row[q] / tot, row[q] / calls)
                                                              for q in range(8) if row[q])))
    if args.inner:
        rh.close()
        return
    if args.sweep and rank == 0:
        def line(tag, el, st):
            ks = {k: round(v[1] / args.steps, 3) for k, v in sorted(st.items(), key=lambda kv: -kv[1][1])[:18]}
            sys.stderr.write("SWEEP " + json.dumps({"opts": tag, "ms_per_step": round(1e3 * el / args.steps, 3), "kernels_ms": ks}) + "\n")
        line("default", elapsed, stats)
        for spec in args.sweep:
            pairs = [kv.split('=') for kv in spec.split(',') if kv]
            for k, v in pairs:
                rh.set_option(int(k), int(v))
            warm(render, 1)
            rh.profile_reset()
            rh.profile(True)
            el = timed(torch, dist, world, dev, render, args.steps)
            rh.profile(False)
            line(spec, el, rh.profile_read())
            for k, v in pairs:
                rh.set_option(int(k), user_opts.get(int(k), OPTION_DEFAULTS.get(int(k), 0)))  # back to the user's value / the option's default
        warm(render, 1)
    # The colour branch runs on the library's second stream (RR_OPT_COLOUR_STREAM, default): the event pairs around ITS kernels
    # then span the time they waited for room beside the other stream's kernels, not their cost.  One more pass of the same
    # step with everything on one in-order stream gives every kernel's own duration (and what the overlap is worth).
    stats_serial, elapsed_serial = None, None
    colour_mode = user_opts.get(hb.RR_OPT_COLOUR_STREAM, 1)
    if not args.no_serial and not args.inner and colour_mode != 0:      # (every rank: the timed loop holds collectives)
        rh.set_option(hb.RR_OPT_COLOUR_STREAM, 0)
        warm(render, 1)
        rh.profile_reset()
        rh.profile(True)
        elapsed_serial = timed(torch, dist, world, dev, render, args.steps)
        rh.profile(False)
        stats_serial = rh.profile_read()
        rh.set_option(hb.RR_OPT_COLOUR_STREAM, user_opts.get(hb.RR_OPT_COLOUR_STREAM, 1))      # (the user's --opt 21=.. stays in force)
        warm(render, 1)
    nb = min(B, batch.n)
    cnts = np.array([rh.batch_counts(i) for i in range(chunks[-1][2])]) if chunks else np.zeros((1, 8), int)

    extras = {}
    single = rank == 0 and world == 1
    # --- variants (N=1): what the headline's conditions hide ------------------------------------
    if is_sim:
        cnt_dev = batch.device_counts()
        batch.mean_drops = float(cnt_dev.mean())
        assert int(cnt_dev.max()) <= batch.cap
    if single and not args.no_variants and not strong and not is_sim:
        var = {}

        def rate(fn, frames, reps=3):
            warm(fn, 1)
            return frames * reps / timed(torch, dist, 1, dev, fn, reps)
        rh.set_option(hb.RR_OPT_DEDUP, 0)
        var["no_tile_sharing"] = {"frames_per_s": rate(render, batch.n), "what": "RR_OPT_DEDUP=0: every drop renders its own raw tile"}
        rh.set_option(hb.RR_OPT_DEDUP, 1)
        nn = min(64, batch.n)
        nz = DeviceBatch(torch, hb, sc, dev, fids[:nn], my_frames[:nn], noise_std=3.0, in_dtype=args.input_dtype)
        nzc = [nz.chunk(hb, 0, nz.n)]
        var["noise_std_3"] = {"frames_per_s": rate(lambda: render(nzc), nz.n),
                              "what": "%d frames with --noise_std 3 --noise_scale 1 (rotations differ per drop: raw tiles stop being shareable)" % nn}
        small = [batch.chunk(hb, a, a + 4) for a in range(0, min(batch.n, 64) - 3, 4)]
        var["calls_of_4"] = {"frames_per_s": rate(lambda: render(small), 4 * len(small)), "what": "library calls of 4 frames"}
        mid = [batch.chunk(hb, a, a + 32) for a in range(0, min(batch.n, 128) - 31, 32)]
        if mid:
            var["calls_of_32"] = {"frames_per_s": rate(lambda: render(mid), 32 * len(mid)), "what": "library calls of 32 frames (the driver's default)"}
        if batch.n >= 256:
            q128 = [batch.chunk(hb, a, a + 128) for a in range(0, batch.n - 127, 128)]
            var["calls_of_128"] = {"frames_per_s": rate(lambda: render(q128), 128 * len(q128)), "what": "library calls of 128 frames (the driver's slot size)"}
        if batch.n >= 512:
            half = [batch.chunk(hb, a, a + 256) for a in range(0, batch.n - 255, 256)]
            var["calls_of_256"] = {"frames_per_s": rate(lambda: render(half), 256 * len(half)), "what": "library calls of 256 frames (the round-3 headline's call size)"}
        del nz, nzc
        other = 'f64' if args.input_dtype == 'f32' else 'f32'
        ob = DeviceBatch(torch, hb, sc, dev, fids[:nn], my_frames[:nn], in_dtype=other)
        obc = [ob.chunk(hb, 0, ob.n)]
        var["inputs_" + other] = {"frames_per_s": rate(lambda: render(obc), ob.n),
                                  "frames_per_s_same_call_size_%s" % args.input_dtype: rate(lambda: render([batch.chunk(hb, 0, nn)]), nn),
                                  "what": "%d frames per call with the image / map inputs resident as %s instead of %s" % (nn, other, args.input_dtype)}
        del ob, obc
        # the reference's own arithmetic (bad_weather.py:397-446 is float64 throughout): float64 inputs, float64 colour branch,
        # float64 compositor, the headline's call size
        try:
            rb = DeviceBatch(torch, hb, sc, dev, fids, my_frames, in_dtype='f64')
            rbc = [rb.chunk(hb, a, min(a + B, rb.n)) for a in range(0, rb.n, B)]
            rh.set_option(hb.RR_OPT_COMPOSITE_F64, 1)
            rh.set_option(hb.RR_OPT_FOV_F32, 0)
            var["reference_arithmetic"] = {"frames_per_s": rate(lambda: render(rbc), rb.n, reps=2),
                                           "what": "%d frames per call, image / map / solid angles resident as float64, RR_OPT_COMPOSITE_F64 1 and RR_OPT_FOV_F32 0: "
                                                   "float64 in every stage, as the reference computes" % min(B, rb.n)}
        except Exception as e:                                 # noqa: BLE001 -- a variant never fails the line
            var["reference_arithmetic"] = {"error": repr(e)}
        finally:
            rh.set_option(hb.RR_OPT_COMPOSITE_F64, user_opts.get(hb.RR_OPT_COMPOSITE_F64, 0))
            rh.set_option(hb.RR_OPT_FOV_F32, user_opts.get(hb.RR_OPT_FOV_F32, 2))
            rb = rbc = None
        warm(render, 1)                      # back to the headline configuration (arena / scratch sized for it)
        extras["variants"] = var

    # --- the fog + environment-map pre-pass (rr_prepass_frames_device); not part of `value` ------------------
    if single and not args.no_prepass and not strong and not is_sim:
        fogmod = importlib.import_module('rain-rendering_amd.common.add_attenuation')
        envmod = importlib.import_module('rain-rendering_amd.common.envmap')
        imgops = importlib.import_module('rain-rendering_amd.common.imgops')
        cs = sc.cam_settings
        consts = fogmod.FogRain(rain_intensity=wl['rate'], focal=cs['focal_mm'] / 1000., f_number=cs['f_number'], angle=90,
                                exposure=cs['exposure_ms'], camera_gain=20).constants()
        rh.set_prepass_kernels(imgops.gaussian_kernel(25, 25), imgops.gaussian_kernel(15, 0))
        we = rh.set_envmap_geometry(H, W, *envmod.EnvironmentMapGenerator(cs['focal_mm'] / 1000., W, H).device_tables(H, W))
        assert we == We
        nb_pre = min(batch.n, 64)
        depth_t = torch.from_numpy((np.linspace(80, 2, H, dtype=np.float32)[:, None] * np.ones((1, W), np.float32))).to(dev)

        def pre_leg(narrow):
            """narrow: the pipeline's types (uint8 image in, float32 fog layer and xyY map out: the float64 results rounded
            once); else float64 in and out (the reference's arrays)."""
            pin = (hb.rr_prepass_in * nb_pre)()
            pout = (hb.rr_prepass_out * nb_pre)()
            keep = []
            for i in range(nb_pre):
                img = np.asarray(batch.host[i % len(batch.host)][0], np.float64)
                o_r = torch.empty((H, W, 3), dtype=torch.float32 if narrow else torch.float64, device=dev)
                o_e = torch.empty((He, We, 3), dtype=torch.float32 if narrow else torch.float64, device=dev)
                i_bg = torch.from_numpy((img * 255).astype(np.uint8) if narrow else img).to(dev)
                keep += [o_r, o_e, i_bg]
                pin[i].H, pin[i].W, pin[i].bg, pin[i].depth, pin[i].depth_f64 = H, W, i_bg.data_ptr(), depth_t.data_ptr(), 0
                pin[i].in_types = hb.RR_IN_BG_U8 if narrow else 0
                pin[i].beta_ext, pin[i].beta_hg, pin[i].irr_num, pin[i].irr_den = [float(v) for v in consts]
                pout[i].rainy_bg, pout[i].env_xyY, pout[i].env_bgr_u8 = o_r.data_ptr(), o_e.data_ptr(), None
                pout[i].out_types = (hb.RR_OUT_RAINY_F32 | hb.RR_OUT_ENV_F32) if narrow else 0

            def run_pre():
                rh._check(rh.lib.rr_prepass_frames_device(rh.h, nb_pre, pin, pout, ctypes.c_void_p(stream)), 'prepass')
            run_pre()
            torch.cuda.synchronize()
            rh.profile_reset()
            rh.profile(True)
            t = timed(torch, dist, 1, dev, run_pre, args.steps)
            rh.profile(False)
            pstats = rh.profile_read()
            del keep
            return {"ms_per_frame": 1e3 * t / args.steps / nb_pre, "frames_per_s": nb_pre * args.steps / t,
                    "kernels_ms_per_call": {k: v[1] / args.steps for k, v in sorted(pstats.items(), key=lambda kv: -kv[1][1])}}
        extras["prepass"] = dict(pre_leg(True), what="fog attenuation + environment map + xyY (rr_prepass_frames_device), %d frames per call, "
                                 "uint8 image in, float32 fog layer + xyY map out (the types rr_pipeline_* hand to the hot path; float64 "
                                 "arithmetic); not in value" % nb_pre, float64_in_and_out=pre_leg(False))
        depth_h = np.ascontiguousarray(depth_t.cpu().numpy())

        # --- the two output files as PNG payloads made on the device: Sub-filtered scanlines (k_png_rows), and -- RR_OPT_PNG_DEFLATE --
        #     their zlib streams (k_pngz: csrc/rr_deflate.h); what the drop-in driver downloads instead of pixels.  Not in `value`.
        nb_png = min(batch.n, 64)
        pfin, pfout, _ = batch.chunk(hb, 0, nb_png)
        row_bytes = H * (1 + 4 * W)
        t_png = torch.zeros((nb_png, 2, row_bytes), dtype=torch.uint8, device=dev)
        for k in range(nb_png):
            pfout[k].rainy_png, pfout[k].mask_png = t_png[k, 0].data_ptr(), t_png[k, 1].data_ptr()
        rh.set_colormap(imgops.viridis_lut())
        png_leg = {}
        for mode, name in ((0, "scanlines"), (1, "zlib_streams")):
            rh.set_option(hb.RR_OPT_PNG_DEFLATE, mode)

            def run_png():
                rh.render_frames_device(pfin, pfout, nb_png, stream)
            run_png()
            assert rh.synchronize()
            rh.profile_reset()
            rh.profile(True)
            t = timed(torch, dist, 1, dev, run_png, args.steps)
            rh.profile(False)
            ks = {k: v[1] / args.steps for k, v in rh.profile_read().items() if k.startswith('k_png')}
            png_leg[name] = {"ms_per_call": 1e3 * t / args.steps, "kernels_ms_per_call": ks}
            if mode == 1:
                heads = t_png[:, :, :8].cpu().numpy()
                coded = (heads[:, :, :4].reshape(-1, 4) == np.frombuffer(b'RRZ1', np.uint8)).all(1)
                lens = heads[:, :, 4:8].copy().view(np.uint32).reshape(nb_png, 2)
                png_leg[name]["files_coded"] = int(coded.sum())
                png_leg[name]["mean_stream_bytes"] = {"image": float(lens[:, 0].mean()), "mask": float(lens[:, 1].mean()), "scanlines": row_bytes}
        rh.set_option(hb.RR_OPT_PNG_DEFLATE, 0)
        extras["png_on_device"] = dict(png_leg, what="%d frames per call with both PNG payloads as outputs; not in value" % nb_png)
        del t_png, pfin, pfout

        # --- host-inclusive (SURVEY 8d's rate: "including H2D of frame inputs and D2H of outputs"): pinned buffers, three
        #     slots in flight (upload | kernels | download overlap); PCIe up (u8 image + f32 depth + drop table), fog +
        #     environment-map pre-pass + hot path on the device, PCIe down (u8 image + int32 mask) -----------------------
        nslot = hb.RR_PIPE_SLOTS
        up = 3 * H * W + 4 * H * W + 112 * batch.mean_drops
        down = 3 * H * W + 4 * H * W

        rh.set_solid_angles(sc.omega)                     # resident: frames pass omega=None

        def host_inclusive(PB, copy_kernels=False, packed=True, prepared=True, depth_u16=True):
            """packed: the frames of a slot back to back in one page-locked block per array (RainHip.host_rows: one copy per
            array and batch) -- else every frame its own allocations (one copy per frame and array).  prepared: descriptor
            arrays built once per slot (what Generator does) -- else rebuilt in Python for every submission.  depth_u16: the
            depth file's uint16 samples travel (RR_DEPTH_U16, what the driver's batch-native route uploads since round 4) --
            else the float32 metres the host made of them (rounds 2-3)."""
            dtd = np.uint16 if depth_u16 else np.float32
            dval = np.rint(depth_h.astype(np.float64) * 256.0).astype(np.uint16) if depth_u16 else depth_h
            up_b = 3 * H * W + (2 if depth_u16 else 4) * H * W + 112 * batch.mean_drops
            rh.set_option(hb.RR_OPT_COPY_KERNELS, int(copy_kernels))           # (0: hipMemcpyAsync; 1: 64 workgroups; n > 1: n workgroups)
            cap = max(len(h_[2]) for h_ in batch.host)
            cap = (cap + 3) // 4 * 4
            slots, blocks = [], []
            for s_ in range(nslot):
                if packed:
                    arrs = [rh.host_rows(PB, shp, dt) for shp, dt in (((H, W, 3), np.uint8), ((H, W), dtd), ((cap,), hb.DROP_DTYPE),
                                                                      ((H, W, 3), np.uint8), ((H, W), np.int32))]
                    blocks += [a[0] for a in arrs]
                    bg8s, deps, drs, ims, mks = [a[1] for a in arrs]
                else:
                    bg8s = [rh.host_array((H, W, 3), np.uint8) for _ in range(PB)]
                    deps = [rh.host_array((H, W), dtd) for _ in range(PB)]
                    drs = [rh.host_array((cap,), hb.DROP_DTYPE) for _ in range(PB)]
                    ims = [rh.host_array((H, W, 3), np.uint8) for _ in range(PB)]
                    mks = [rh.host_array((H, W), np.int32) for _ in range(PB)]
                    blocks += bg8s + deps + drs + ims + mks
                frs, outs, nds = [], [], []
                for k in range(PB):
                    i = (s_ * PB + k) % batch.n
                    bg8s[k][...] = (batch.host[i][0] * 255).astype(np.uint8)
                    deps[k][...] = dval
                    nd = len(batch.host[i][2])
                    drs[k][:nd] = batch.host[i][2]
                    nds.append(nd)
                    frs.append(dict(bg_u8=bg8s[k], depth=deps[k], fog=consts, omega=None, drops=drs[k]))
                    outs.append(dict(image_u8=ims[k], mask_i32=mks[k]))
                prep = rh.pipeline_prepare(frs, outs)
                for k, nd in enumerate(nds):
                    prep.set_drop_count(k, nd)
                slots.append((frs, outs, nds, prep))

            def submit(s_):
                frs, outs, nds, prep = slots[s_]
                if prepared:
                    rh.pipeline_submit_prepared(s_, prep)
                else:
                    rh.pipeline_submit(s_, [dict(fr, drops=fr['drops'][:nd]) for fr, nd in zip(frs, nds)], outs)

            def pipe(rounds):
                done = 0
                for r in range(rounds + nslot):
                    s_ = r % nslot
                    if r >= nslot:
                        while not rh.pipeline_wait(s_):
                            submit(s_)
                        done += PB
                    if r < rounds:
                        submit(s_)
                return done
            pipe(nslot)                                   # warm-up: staging buffers, arena
            rounds = max(4 * nslot, (2048 + PB - 1) // PB)  # >= 12 batches and >= 2048 frames: the ramp-up of the three slots is a small part
            h0 = time.perf_counter()
            done = pipe(rounds)
            h1 = time.perf_counter()
            slots = None
            for a_ in blocks:
                rh.host_free(a_)
            rh.set_option(hb.RR_OPT_COPY_KERNELS, 0)
            return {"frames_per_s": done / (h1 - h0), "ms_per_frame": 1e3 * (h1 - h0) / done, "frames_per_slot": PB, "frames_timed": done,
                    "copies": ("one copy kernel per direction and batch (RR_OPT_COPY_KERNELS %d)" % int(copy_kernels) if copy_kernels else
                               ("hipMemcpyAsync, one per array and batch (frames back to back in one page-locked block per array)" if packed
                                else "hipMemcpyAsync, one per frame and array (separate allocations)")),
                    "descriptors": "prepared once per slot" if prepared else "rebuilt in Python for every submission",
                    "depth_upload": "uint16 samples of the depth file (RR_DEPTH_U16)" if depth_u16 else "float32 metres",
                    "pcie_bytes_per_frame": {"up": up_b, "down": down},
                    "pcie_GBps": {"up": up_b * done / (h1 - h0) / 1e9, "down": down * done / (h1 - h0) / 1e9}}
        PB = max(1, min(args.pipe_batch, batch.n))
        hi = host_inclusive(PB)
        hi["what"] = ("rr_pipeline_submit/wait, %d slots x %d frames, pinned host buffers (rr_host_alloc): PCIe up (u8 image + u16 depth samples "
                      "+ drop table), fog + environment-map pre-pass + hot path on the device, PCIe down (u8 image + int32 mask); PNG codec excluded"
                      % (nslot, PB))
        extras["host_inclusive"] = hi
        if args.copy_sweep:
            extras["host_inclusive_copy_sweep"] = {"blocks_%s" % b: host_inclusive(PB, copy_kernels=int(b))["frames_per_s"] for b in args.copy_sweep.split(',')}
        if not args.no_variants:
            extras["host_inclusive_variants"] = {"slots_of_32": host_inclusive(min(32, batch.n)),
                                                 "slots_of_256": host_inclusive(min(256, batch.n)),      # (with 512 frames in all a slot size of 256 is two batches: the third slot stays empty and nothing overlaps the last download)
                                                 "copy_kernels": host_inclusive(PB, copy_kernels=True),
                                                 "depth_as_float32": host_inclusive(PB, depth_u16=False)}
            if args.more_variants:
                extras["host_inclusive_variants"].update({"slots_of_64": host_inclusive(min(64, batch.n)),
                                                          "separate_allocations": host_inclusive(PB, packed=False),
                                                          "descriptors_rebuilt": host_inclusive(PB, prepared=False)})
        warm(render, 1)

    if rank == 0:
        frames_step = args.total_frames if strong else B * world
        fps = frames_step * args.steps / elapsed
        per_launch = {k: v[1] / v[0] for k, v in stats.items()}
        # the scopes launched on the library's second stream: mode 1 the FOV chain, mode 2 the bookkeeping chain + k_colour
        COLOUR = (('k_fov_spans', 'k_fov_sums', 'k_fov_poly', 'k_env_prefix', 'k_fov_sums_general') if colour_mode == 1 else
                  ('k_plan', 'k_scan', 'k_dedup', 'k_lists', 'k_colour'))
        overlapped = {k: v for k, v in per_launch.items() if k in COLOUR} if stats_serial is not None else {}
        per_main = {k: v for k, v in per_launch.items() if k not in overlapped}
        dom_name = max(per_main, key=per_main.get)
        avg_ms = per_main[dom_name]
        per_serial = {k: v[1] / v[0] for k, v in stats_serial.items()} if stats_serial is not None else None
        alg = algorithmic_bytes(H, W, He, We, batch.mean_drops) * nb
        achieved = alg / (avg_ms * 1e-3) / 1e9
        traffic, traffic_how = None, "not measured (--no-traffic, N>1 or strong scaling)"
        if single and not args.no_traffic and not strong:
            parts = {'k_env_prefix': ['k_env_prefix', 'k_env_consts']}.get(dom_name, [dom_name])
            traffic, traffic_how = measure_traffic(args, parts, scene_dir=tmp)
        chain_ms = sum((per_serial or per_launch).values())         # every kernel's own duration: the one-stream pass when there is one
        compute = None
        if not is_sim and not strong:
            try:                       # float64 operation model of the blur kernels (the dominant ones), a few frames scaled to the call
                acc = {}
                nfr = min(8, batch.n)
                for k in range(nfr):       # per-frame statuses live in the output tensors of the batch
                    st = batch.status_tensors[k].cpu().numpy()
                    m = blur_flop_model(batch.host[k][2], st[:len(batch.host[k][2])], sc.cam, H, W)
                    for kk, vv in m.items():
                        acc[kk] = acc.get(kk, 0.0) + vv
                scale = nb / float(nfr)
                compute = {"what": "float64 operations of the defocus blur per library call: 1 + 3 r per filter output of radius r (plan "
                                   "geometry restated with numpy from the drop records, %d frames scaled to %d); no FMA contraction on the "
                                   "alpha path, so the attainable rate is HALF the peak quoted" % (nfr, nb),
                           "peak_TFLOPs": F64_VECTOR_PEAK_TFLOPS, "kernels": {}}
                for kn in ("k_blur_fused", "k_blur_small", "k_blur_rows"):
                    if kn in per_launch and acc.get(kn, 0.0) > 0:
                        tf = acc[kn] * scale / (per_launch[kn] * 1e-3) / 1e12
                        compute["kernels"][kn] = {"flops_per_launch": acc[kn] * scale, "ms": per_launch[kn], "achieved_TFLOPs": tf,
                                                  "frac_of_f64_vector_peak": tf / F64_VECTOR_PEAK_TFLOPS,
                                                  "frac_of_rate_without_fma": 2.0 * tf / F64_VECTOR_PEAK_TFLOPS}
            except Exception as e:                                  # noqa: BLE001 -- reported, never fatal
                compute = {"error": repr(e)}
        valu, valu_how = None, "not measured (--no-traffic, N>1 or strong scaling)"
        if single and not args.no_traffic and not strong:
            valu, valu_how = measure_valu(args, scene_dir=tmp)
        # which roof is nearer for the dominant kernel: its share of the HBM peak (measured traffic, else algorithmic bytes) or of the
        # float64 vector rate (operation model for the blur kernels, else the PMC pass' issue-slot share)
        hbm_frac = ((traffic if traffic else alg) / (avg_ms * 1e-3) / 1e9) / HBM_PEAK_GBS if (traffic and dom_name) else achieved / HBM_PEAK_GBS
        valu_frac = None
        dom_valu = next((v for k, v in (valu or {}).items() if scope_of(k) == dom_name), None)      # (the longest-running kernel of the scope comes first)
        if compute and dom_name in (compute.get("kernels") or {}):
            valu_frac = compute["kernels"][dom_name]["frac_of_rate_without_fma"]
        elif valu and dom_valu:
            valu_frac = dom_valu["valu_util"]
        nearest = "hbm" if (valu_frac is None or (traffic and hbm_frac >= valu_frac)) else "valu-f64"
        # neither roof within a factor of two: what bounds the kernel is the latency of its dependent memory / LDS phases at the
        # occupancy its registers and LDS tiles allow -- say so instead of naming a roof it is far from
        bound = nearest if max(hbm_frac, valu_frac or 0.0) >= 0.5 else "latency"
        bound_detail = ("dominant kernel %s: %.0f %% of the HBM peak by its measured traffic%s; neither roof is reached -- the kernel waits on "
                        "dependent loads and LDS phases (DESIGN.md, phase clocks); no MFMA: the path has no dense contraction"
                        % (dom_name, 100.0 * hbm_frac if traffic else float('nan'),
                           (", %.0f %% of the float64 vector rate attainable without FMA contraction" % (100.0 * valu_frac)) if valu_frac is not None else ""))
        out = {
            "metric": wl['metric'],
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f64 alpha/mask, f32 colour", "data": "synthetic",
            "config": {"workload": ("%s shape %dx%d, %d mm/hr (%d streaks/frame simulated, %.0f after the frame filter), precomputed "
                                    "particles; BASELINE.json %s" % (wl['cam'], W, H, wl['rate'], N, batch.mean_drops, wl['cfg'])) if not is_sim else
                                   ("%s shape %dx%d, %d mm/hr, IN-KERNEL particle simulation inside the timed step (%.0f particles/frame "
                                    "simulated on the device, %.0f drops after the frame filter; no XML, no host drop table); BASELINE.json %s"
                                    % (wl['cam'], W, H, wl['rate'], float(sims['n_particles'].mean()), batch.mean_drops, wl['cfg'])),
                       "frames_per_call": nb, "frames_per_step": frames_step, "envmap": "%dx%d" % (We, He),
                       "inputs_in_hbm": ("float32 image + float32 xyY map / solid angles (rr_frame_in.in_types)" if args.input_dtype == 'f32' else
                                         "float64 image, map and solid angles (the reference's arrays)") + ", 112-byte drop records",
                       "parallelism": "frames sharded round-robin, dp%d; one RCCL broadcast of the streak DB" % world,
                       "raw_tiles_last_call": {"rotate_resize": int(cnts[:, 0].sum()), "bicubic_warp": int(cnts[:, 5].sum()),
                                               "generic": int(cnts[:, 1].sum()), "shared_bit_identical": int(cnts[:, 7].sum())},
                       "raw_tile_share": float(cnts[:, 7].sum()) / max(1.0, float(cnts[:, 0].sum() + cnts[:, 1].sum() + cnts[:, 5].sum() + cnts[:, 7].sum())),
                       "blur_last_call": {"fused_items": int(cnts[:, 2].sum()), "wave_per_drop": int(cnts[:, 4].sum()),
                                          "two_pass": int(cnts[:, 3].sum())}},
            "roofline": {"bound": bound, "nearest_roof": nearest, "bound_detail": bound_detail, "compute": compute, "kernel": dom_name, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "chain_frac_of_hbm_peak": alg / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,      # the same bytes over the SUM of all kernels of the step
                         "traffic": traffic, "traffic_source": traffic_how,
                         "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": alg,
                         "definition": "algorithmic bytes of the frames of one launch (27*H*W + 16*He*We + 64*N each, SURVEY 8d) / "
                                       "average launch time of the slowest kernel of the chain (HIP events on the launch stream)"},
            "chain": {"ms_per_call_sum_of_kernels": chain_ms, "algorithmic_GBps": alg / (chain_ms * 1e-3) / 1e9,
                      "frac_of_hbm_peak": alg / (chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "kernels_ms_per_call": {k: v for k, v in sorted(per_main.items(), key=lambda kv: -kv[1])},
            "overlap": None if per_serial is None else {
                "what": "the FOV chain (polygons, spans, sums over the environment map) runs on a second stream of the library beside plan .. "
                        "tiles .. blur (RR_OPT_COLOUR_STREAM 1).  kernels_ms_per_call: the caller's stream's kernels in the timed region; "
                        "colour_branch_ms_between_events: the second stream's kernels between their event pairs -- including the time their "
                        "workgroups waited for room on the CUs, not a cost; kernels_ms_per_call_one_stream: the same step with the option 0, "
                        "every kernel alone on the device",
                "ms_per_step": 1e3 * elapsed / args.steps, "ms_per_step_one_stream": 1e3 * elapsed_serial / args.steps,
                "colour_branch_ms_between_events": {k: v for k, v in sorted(overlapped.items(), key=lambda kv: -kv[1])},
                "kernels_ms_per_call_one_stream": {k: v for k, v in sorted(per_serial.items(), key=lambda kv: -kv[1])}},
            "valu": {"what": valu_how, "dominant_kernel": dom_valu["valu_util"] if dom_valu else None,
                     "per_kernel": {k: {"valu_util": round(v["valu_util"], 4), "waiting": round(v["waiting"], 4) if v["waiting"] is not None else None}
                                    for k, v in sorted((valu or {}).items(), key=lambda kv: -kv[1]["valu_util"])} if valu else None},
        }
        out.update(extras)
        if "host_inclusive" in extras:                         # SURVEY 8(d)'s own rate (H2D of the inputs and D2H of the outputs included), at the top level
            out["host_inclusive_frames_per_s"] = extras["host_inclusive"].get("frames_per_s")
        if single and not args.no_driver and not strong and not is_sim and args.workload == 'kitti100':
            out["driver_end_to_end"] = driver_end_to_end(1024, args.pipe_batch)
        if not args.no_cpu_baseline and single and not strong and not is_sim:
            out["cpu_baseline"] = cpu_baseline(sc, batch.host, W, H, args.cpu_sample_drops, os.cpu_count() or 1)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    rh.close()


if __name__ == '__main__':
    main()
