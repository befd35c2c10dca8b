"""Seeded synthetic inputs in the reference's own on-disk formats (the reference ships no
data: rainstreakdb, particle files and images must be downloaded / simulated).

  * write_streak_db      <root>/env_light_database/{size32/cv<k>_osc<j>.png, txt/normalized_env_max.txt}
                         (16-bit gray PNGs; layout read by DBManager.load_streak_database,
                         reference bad_weather.py:108-146)
  * simulate_particles   a physically motivated stand-in for the external AHLSimulation
                         binary (reference tools/simulation.py drives it; no source exists),
                         emitting the XML schema load_streaks_from_xml parses
                         (reference bad_weather.py:192-211)
  * make_frame / make_envmap   image-like float frames and lat-long environment maps

Fixed drop counts per fall rate are a synthetic choice: real counts are unknown.
"""
import os

import numpy as np

DROPS_PER_RATE = {1: 128, 5: 512, 25: 2048, 50: 4096, 100: 8192, 200: 16384}

# (width, heights) of the synthetic textures: 5 camera views x 10 oscillations; the 5 distinct
# w/h ratios are what take_drop_texture's bucket logic needs (bad_weather.py:250-265)
TEX_W = 32
TEX_H = (320, 229, 160, 114, 80)


def envmap_width(focal_mm, imW):
    """Width of the lat-long map EnvironmentMapGenerator.generate_map builds for an
    imW-wide frame (reference bad_weather.py:712,730-749,791): cylinder + two flipped halves."""
    focal = int((focal_mm / 12.7) * imW)
    cx = int(imW // 2)
    max_x = round(focal * np.arctan(cx / focal) + cx)
    min_x = round(focal * np.arctan(-cx / focal) + cx)
    cyl_w = int(max_x - min_x) + 1
    return cyl_w + 2 * int(cyl_w / 2)


def _box_blur(a, k):
    from scipy.ndimage import uniform_filter
    size = (k, k) + (1,) * (a.ndim - 2)
    return uniform_filter(a, size=size, mode='nearest')


def make_frame(i, H, W):
    """Image-like float64 BGR frame in [0,1]: low-pass filtered uniform noise."""
    rng = np.random.RandomState(1000 + i)
    a = _box_blur(rng.rand(H, W, 3), 9)
    a = (a - a.min()) / (a.max() - a.min())
    return np.ascontiguousarray(0.1 + 0.7 * a)


def make_envmap(i, He, We):
    """BGR environment map in (0,1]."""
    rng = np.random.RandomState(2000 + i)
    a = _box_blur(rng.rand(He, We, 3), 15)
    a = (a - a.min()) / (a.max() - a.min())
    return np.ascontiguousarray(0.05 + 0.9 * a)


def make_textures(seed=7, tex_heights=None, tex_width=None):
    """50 uint16 gray streak images: a Gaussian ridge along the streak modulated by the
    drop's shape oscillation (Garg & Nayar style appearance), brighter at the ends."""
    rng = np.random.RandomState(seed)
    out = []
    tw = tex_width or TEX_W
    for v, h in enumerate(tex_heights or TEX_H):
        y = (np.arange(h) + 0.5) / h
        x = (np.arange(tw) + 0.5) / tw - 0.5
        for osc in range(10):
            amp = 0.05 + 0.03 * osc
            freq = 1.5 + 0.7 * osc + 0.2 * v
            centre = amp * np.sin(2 * np.pi * freq * y + rng.uniform(0, 2 * np.pi))
            width = 0.10 + 0.04 * np.cos(2 * np.pi * freq * y) ** 2
            prof = np.exp(-0.5 * ((x[None, :] - centre[:, None]) / width[:, None]) ** 2)
            env = np.sin(np.pi * y) ** 0.5
            img = prof * env[:, None] * (0.75 + 0.25 * np.cos(4 * np.pi * freq * y))[:, None]
            img = img / img.max()
            out.append((v, osc, np.round(img * 65535).astype(np.uint16)))
    return out


def write_streak_db(root, seed=7, tex_heights=None, tex_width=None):
    """Writes the rainstreakdb layout under `root`; returns (texture_dir, norm_file)."""
    from PIL import Image
    tex_dir = os.path.join(root, 'env_light_database', 'size32')
    txt_dir = os.path.join(root, 'env_light_database', 'txt')
    os.makedirs(tex_dir, exist_ok=True)
    os.makedirs(txt_dir, exist_ok=True)
    rng = np.random.RandomState(seed + 1)
    coeffs = {}
    for v, osc, img in make_textures(seed, tex_heights, tex_width):
        Image.fromarray(img).save(os.path.join(tex_dir, 'cv%d_osc%d.png' % (v, osc)))
        coeffs.setdefault(v, []).append(0.3 + 0.7 * rng.rand())
    norm = os.path.join(txt_dir, 'normalized_env_max.txt')
    with open(norm, 'w') as fh:
        for v in sorted(coeffs):
            fh.write('cv%d\n' % v)
            fh.write(''.join('%.6f ' % c for c in coeffs[v]) + '\n')      # trailing space: bad_weather.py:129
    return tex_dir, norm


def simulate_particles(n_frames, n_drops, W, H, focal_mm=6.0, pix_um=4.65, exposure_ms=2.0, seed0=3000,
                       cam_speed_kmh=30.0, far_fraction=0.02):
    """Per frame: n_drops streaks as dicts of XML attribute values.

    Camera at the origin looking along -z (simulator convention; the loader negates z,
    reference bad_weather.py:223-224), x right, y up, image origin bottom-left (the loader
    flips y, bad_weather.py:221-222).  A drop of diameter D at depth z shows a streak of
    width D*f/z pixels; the image-width mix (60% Small, 25% Medium, 15% Big) is drawn
    first and the depth follows from it; `far_fraction` of the drops sit beyond the 10 m
    rendering sphere to exercise the skip path."""
    fpx = focal_mm * 1e-3 / (pix_um * 1e-6)
    t = exposure_ms * 1e-3
    frames = []
    for fi in range(n_frames):
        rng = np.random.RandomState(seed0 + fi)
        drops = []
        for k in range(n_drops):
            u = rng.rand()
            if u < 0.60:
                iw = rng.uniform(1.05, 1.95)
            elif u < 0.85:
                iw = rng.uniform(2.05, 3.95)
            else:
                iw = 4.0 + rng.exponential(2.5)
            D = rng.uniform(0.5e-3, 5e-3) if iw < 4 else rng.uniform(1.5e-3, 5e-3)
            depth = D * fpx / iw
            if rng.rand() < far_fraction:
                depth = rng.uniform(10.5, 15.0)
            depth = min(max(depth, 0.25), 15.0)
            # uniform position in the image, slightly beyond the borders
            px = rng.uniform(-0.03 * W, 1.03 * W)
            py = rng.uniform(-0.05 * H, 1.05 * H)              # from the bottom
            X = (px - W / 2) * depth / fpx
            Y = (py - H / 2) * depth / fpx
            Z = -depth
            vfall = 9.65 - 10.3 * np.exp(-0.6 * D * 1e3)        # terminal velocity (Atlas et al.)
            wind = rng.normal(0.0, 1.0)
            vc = cam_speed_kmh / 3.6
            X2, Y2, Z2 = X + wind * t, Y - vfall * t, Z + vc * t
            depth2 = max(-Z2, 0.05)
            px2 = W / 2 + fpx * X2 / depth2
            py2 = H / 2 + fpx * Y2 / depth2
            iw1 = D * fpx / depth if depth < 10.0 else iw
            iw2 = D * fpx / depth2 if depth < 10.0 else iw
            drops.append(dict(pid=k, wp1=(X, Y, Z), wp2=(X2, Y2, Z2), wd1=D, wd2=D,
                              ip1=(px, py), ip2=(px2, py2), iw1=iw1, iw2=iw2))
        frames.append(dict(id=fi, t=int(round(exposure_ms * 1000)), d=fi * 100000, drops=drops))
    return frames


def write_particles_xml(path, frames):
    """The schema load_streaks_from_xml reads (reference bad_weather.py:192-211): the root's
    children are frames (attributes id, t, d, rs); their children are drops (pid, wp1, wp2,
    wd1, wd2, ip1, ip2, iw1, iw2); vectors are "(a;b;c)".  Numbers are written with 17 significant
    digits (they parse back to the same doubles), formatted a frame at a time."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'w') as fh:
        fh.write('<?xml version="1.0" ?>\n<simulation>\n')
        for fr in frames:
            fh.write(_format_frame(fr))
        fh.write('</simulation>\n')
    return path


_STREAK_FMT = ('    <streak pid="%d" wp1="(%.17g;%.17g;%.17g)" wp2="(%.17g;%.17g;%.17g)" wd1="%.17g" wd2="%.17g" '
               'ip1="(%.17g;%.17g)" ip2="(%.17g;%.17g)" iw1="%.17g" iw2="%.17g"/>')


def _format_frame(fr):
    ds = fr['drops']
    out = ['  <frame id="%d" t="%d" d="%d" rs="%d">\n' % (fr['id'], fr['t'], fr['d'], len(ds))]
    if ds:
        rows = [(d['pid'],) + tuple(float(c) for c in d['wp1']) + tuple(float(c) for c in d['wp2']) +
                (float(d['wd1']), float(d['wd2'])) + tuple(float(c) for c in d['ip1']) + tuple(float(c) for c in d['ip2']) +
                (float(d['iw1']), float(d['iw2'])) for d in ds]
        out.append('\n'.join(_STREAK_FMT % r for r in rows))
        out.append('\n')
    out.append('  </frame>\n')
    return ''.join(out)


def _frame_xml(args):
    """One simulated frame as the text write_particles_xml would write for it (a pool worker)."""
    fi, n_drops, W, H, focal_mm, pix_um, exposure_ms, seed0, far_fraction = args
    fr = simulate_particles(1, n_drops, W, H, focal_mm, pix_um, exposure_ms, seed0=seed0 + fi, far_fraction=far_fraction)[0]
    fr['id'], fr['d'] = fi, fi * 100000
    return _format_frame(fr)


def simulate_to_xml(path, n_frames, n_drops, W, H, focal_mm=6.0, pix_um=4.65, exposure_ms=2.0, seed0=3000, far_fraction=0.02, workers=None):
    """simulate_particles + write_particles_xml, the frames split over worker PROCESSES (every frame has its own seed): the
    file is byte for byte what the two calls write -- bench.py's 256-frame scenes take a minute of one core otherwise.
    The workers are plain child interpreters running this file (no fork of a process that may hold a GPU context, no
    re-import of the caller's main module); any failure falls back to the sequential path."""
    import subprocess
    import sys
    workers = workers if workers is not None else max(1, min(8, (os.cpu_count() or 1)))
    os.makedirs(os.path.dirname(path), exist_ok=True)
    # plain Python numbers whatever the caller passed (a numpy scalar's repr is not a literal the worker can parse)
    common = (int(n_drops), int(W), int(H), float(focal_mm), float(pix_um), float(exposure_ms), int(seed0), float(far_fraction))
    parts = []
    if workers > 1 and n_frames >= 16:
        import uuid
        tag = '%d_%s' % (os.getpid(), uuid.uuid4().hex[:8])       # two processes building the same scene do not share part files
        per = (n_frames + workers - 1) // workers
        procs = []
        for k in range(workers):
            a, b = k * per, min(n_frames, (k + 1) * per)
            if a >= b:
                break
            part = '%s.%s.part%d' % (path, tag, k)
            cmd = [sys.executable, os.path.abspath(__file__), part, str(a), str(b)] + [repr(v) for v in common]
            procs.append((subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE), part))
        ok = True
        for pr, part in procs:
            try:
                err = pr.communicate(timeout=900)[1]
                if pr.returncode != 0:
                    ok = False
                    sys.stderr.write("simulate_to_xml: worker for %s failed (%d): %s\n" % (part, pr.returncode, err.decode(errors='replace')[-500:]))
            except subprocess.TimeoutExpired:
                pr.kill()
                pr.communicate()                                    # reap the worker, close its pipe
                ok = False
            parts.append(part)
        if not ok:
            sys.stderr.write("simulate_to_xml: falling back to the sequential path\n")
            for part in parts:
                if os.path.exists(part):
                    os.remove(part)
            parts = []
    tmp_path = '%s.%d.tmp' % (path, os.getpid())               # (the finished file appears under its name at once)
    try:
        with open(tmp_path, 'w') as fh:
            fh.write('<?xml version="1.0" ?>\n<simulation>\n')
            if parts:
                for part in parts:
                    with open(part) as src:
                        while True:
                            blk = src.read(1 << 24)
                            if not blk:
                                break
                            fh.write(blk)
                    os.remove(part)
            else:
                for fi in range(n_frames):
                    fh.write(_frame_xml((fi,) + common))
            fh.write('</simulation>\n')
        os.replace(tmp_path, path)
    finally:
        if os.path.exists(tmp_path):                             # an exception on the way: no half-written file is left behind
            os.remove(tmp_path)
    return path


def write_dataset(root, dataset, sequence, n_frames, H, W, depth_m=20.0):
    """A minimal on-disk dataset in the layout config/<dataset>.py expects:
    <root>/<dataset>/<sequence>/{image_2/*.png, image_2/depth/*.png, calib/*.txt} (kitti-like).
    Images are 8-bit PNGs, depth 16-bit PNGs (metres*256, generator.py:365)."""
    from PIL import Image
    img_dir = os.path.join(root, dataset, sequence, 'image_2')
    dep_dir = os.path.join(img_dir, 'depth')
    cal_dir = os.path.join(root, dataset, sequence, 'calib')
    for d in (img_dir, dep_dir, cal_dir):
        os.makedirs(d, exist_ok=True)
    for i in range(n_frames):
        bgr = make_frame(i, H, W)
        Image.fromarray((bgr[..., ::-1] * 255).astype(np.uint8)).save(os.path.join(img_dir, '%06d.png' % i))
        ramp = np.linspace(80.0, 2.0, H)[:, None] * np.ones((1, W)) if depth_m is None else np.full((H, W), depth_m)
        Image.fromarray(np.round(ramp * 256).astype(np.uint16)).save(os.path.join(dep_dir, '%06d.png' % i))
        with open(os.path.join(cal_dir, '%06d.txt' % i), 'w') as fh:
            fh.write('P2: ' + ' '.join(['0'] * 12) + '\n')
    return img_dir, dep_dir


if __name__ == '__main__':        # worker of simulate_to_xml: frames [a, b) as XML text into a part file
    import sys
    _part, _a, _b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    _n_drops, _W, _H = int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    _rest = [float(v) for v in sys.argv[7:10]] + [int(sys.argv[10]), float(sys.argv[11])]
    with open(_part, 'w') as _fh:
        for _fi in range(_a, _b):
            _fh.write(_frame_xml((_fi, _n_drops, _W, _H) + tuple(_rest)))
