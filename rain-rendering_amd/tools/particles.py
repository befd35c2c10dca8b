"""Rain-particle generator: the stand-in for the reference's external simulator, on the host AND on the device.

The reference drives a closed-source binary (``3rdparty/weather-particle-simulator/.../AHLSimulation``) through
``tools/simulation.py`` / ``tools/particles_simulation.py`` with the settings of ``common/db.py:41-70``
(``cam_hz``, ``cam_CCD_WH``, ``cam_CCD_pixsize``, ``cam_focal``, ``cam_exposure``, ``sim_mode``, ``sim_steps``
{``cam_motion``, ``cam_exposure``, ``cam_focal``, ``rain_fallrate``}, ``sim_duration``) and reads its output back as
``<particles>/<dataset>/<sequence>/rain/<N>mm/*_camera0.xml`` (schema: bad_weather.py:192-211).  There is no source
for that binary, so there is nothing to compare against bit for bit: this is a physically motivated generator, validated
statistically (tests/test_particles.py), that takes the SAME settings and emits the SAME schema.

The model is stated twice, with identical bits:
  * HERE, in numpy (``generate``): the product's host path -- it writes the XML file the reference's loader reads when a
    simulation is missing (``simulate``, main.py) -- and the definition the device is tested against;
  * in ``csrc/rr_particles.h`` (``k_particles`` in rainhip.hip): BASELINE.json configs[4], "in-kernel particle simulation
    (no XML)": the library generates the particles, applies the loader's derived fields and the frame filter, makes the
    renderer's per-drop random draws and leaves ``rr_drop[]`` records in HBM (``rr_generate_drops_device``, or
    ``rr_frame_in.sim`` in the host-pointer entry points).  ``sim_frames`` / ``diameter_tables`` below describe a run to
    the library; ``expected_records`` is what it must produce (tests/test_gpu_particles.py: bit for bit).
Arithmetic shared by both: IEEE double with the evaluation order spelled out, + - * / sqrt rint only (exp through
``det_exp``: the same operations on every machine), random numbers from the counter-based Philox4x32-10 -- particle i of
frame k is a pure function of (seed, k, i), so any lane of any GPU can make it.

Model (per camera frame, camera at the origin looking along -z, x right, y up, image origin bottom-left -- the
conventions the loader undoes, bad_weather.py:221-224):
  * drop diameters follow Marshall & Palmer (1948): N(D) = N0 exp(-Lambda D), N0 = 8000 m^-3 mm^-1,
    Lambda = 4.1 R^-0.21 mm^-1 for a fall rate R in mm/hr, D in [0.5, 6] mm;
  * only drops that can appear at least `min_px` wide are simulated: depth z <= D f / (pixel * min_px) (and <= z_far),
    uniformly in the viewing frustum up to that depth -- the expected count is the integral of N(D) over that volume,
    the actual count of a frame is Poisson distributed (drawn by the host: one number per frame);
  * a drop falls at its terminal velocity v(D) = 9.65 - 10.3 exp(-0.6 D) m/s (Atlas et al. 1973), drifts with a
    horizontal wind (bell-shaped: a centred sum of four uniforms scaled to `wind_sigma`) and approaches the camera at the
    vehicle's speed (``sim_steps['cam_motion']``, km/h); the streak is the path covered during the exposure, both ends
    projected through the pinhole camera;
  * image widths are D f / (pixel z) at either end.
"""
import os

import numpy as np

from ..common.bad_weather import PARTICLE_DTYPE, PARTICLE_FRAME_DTYPE

N0 = 8000.0                      # m^-3 mm^-1
D_MIN, D_MAX = 0.5, 6.0          # mm
N_GRID = 512                     # entries of the diameter table


def mp_lambda(fallrate):
    """Marshall-Palmer slope, mm^-1."""
    return 4.1 * float(fallrate) ** -0.21


def det_exp(x):
    """exp(x) for -700 < x <= 0 from + - * / (and an exact power of two) only: identical bits in numpy, g++ and on gfx950
    (csrc/rr_device.h det_exp is the same sequence of operations)."""
    x = np.asarray(x, np.float64)
    k = np.rint(x * 1.44269504088896338700e+00)
    r = (x - k * 6.93147180369123816490e-01) - k * 1.90821492927058770002e-10
    p = np.full_like(r, 1.0 / 6227020800.0)
    for c in (479001600.0, 39916800.0, 3628800.0, 362880.0, 40320.0, 5040.0, 720.0, 120.0, 24.0, 6.0, 2.0):
        p = p * r + 1.0 / c
    p = p * r + 1.0
    p = p * r + 1.0
    return np.ldexp(p, k.astype(np.int64))


def terminal_velocity(d_mm):
    """m/s (Atlas, Srivastava & Sekhon 1973)."""
    return 9.65 - 10.3 * det_exp(-0.6 * np.asarray(d_mm, np.float64))


# ---- Philox4x32-10 (Salmon, Moraes, Dror & Shaw, SC'11) ---------------------------------------------------------------
_M0, _M1, _W0, _W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32(c0, c1, c2, c3, key0, key1):
    """Four uint32 words per counter (c0..c3 broadcast against each other); the key is two 32-bit integers."""
    c = [np.asarray(v, np.uint64) & _MASK for v in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = int(key0) & 0xFFFFFFFF, int(key1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = np.uint64(_M0) * c[0]
        p1 = np.uint64(_M1) * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & _MASK, p1 >> np.uint64(32), p1 & _MASK
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return [v.astype(np.uint32) for v in c]


def unit32(w):
    """A 32-bit word as a number strictly inside (0, 1): (w + 1/2) / 2^32, exact in double."""
    return (np.asarray(w, np.float64) + 0.5) * (1.0 / 4294967296.0)


class FrameCamera:
    def __init__(self, options, step):
        """The camera of simulation step `step`: `options` is what common.db.sim() returns under "options"."""
        steps = options.get("sim_steps", {}) or {}

        def stepped(key, default):
            v = steps.get(key)
            if v is None or len(v) == 0:
                return default
            return float(v[min(step, len(v) - 1)])           # a parameter stays applied unless later changed (db.py)
        self.W, self.H = (int(v) for v in options["cam_CCD_WH"])
        self.pix = options["cam_CCD_pixsize"] * 1e-6
        self.focal = stepped("cam_focal", options["cam_focal"]) * 1e-3
        self.exposure = stepped("cam_exposure", options["cam_exposure"]) * 1e-3
        self.speed = stepped("cam_motion", 0.0) / 3.6         # km/h -> m/s
        self.fpx = self.focal / self.pix
        self.hz = options["cam_hz"]


def expected_count(cam, fallrate, min_px=1.0, z_far=15.0, margin=0.05, n_grid=N_GRID):
    """(expected visible drops per frame, diameter grid, its sampling CDF, z_max per diameter)."""
    lam = mp_lambda(fallrate)
    d = np.linspace(D_MIN, D_MAX, n_grid)
    z_max = np.minimum(d * 1e-3 * cam.fpx / min_px, z_far)
    area = (1 + 2 * margin) ** 2 * cam.W * cam.H / cam.fpx ** 2          # frustum cross-section at unit depth
    dens = N0 * det_exp(-lam * d) * area * z_max ** 3 / 3.0              # drops per mm of diameter
    cdf = np.concatenate([[0.0], np.cumsum(0.5 * (dens[1:] + dens[:-1]) * np.diff(d))])
    total = float(cdf[-1])
    cdf = cdf / cdf[-1]
    cdf[-1] = 1.0
    return total, d, cdf, z_max


def sample_diameter(dgrid, cdf, u):
    """Inverse-CDF sample: the last j with cdf[j] <= u, linear inside the cell (rr_particles.h sample_diameter)."""
    j = np.minimum(np.searchsorted(cdf, u, side='right') - 1, len(cdf) - 2)
    slope = (dgrid[j + 1] - dgrid[j]) / (cdf[j + 1] - cdf[j])
    return dgrid[j] + (u - cdf[j]) * slope


def _frame_settings(options, fallrate, k, min_px, z_far, margin):
    steps = options.get("sim_steps", {}) or {}
    cam = FrameCamera(options, k)
    rates = steps.get("rain_fallrate", ())
    rate = float(rates[min(k, len(rates) - 1)]) if len(rates) else float(fallrate)
    return cam, rate


def _key(seed):
    seed = int(seed)
    return seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF


def frame_counts(options, fallrate, n_frames, seed=0, min_px=1.0, z_far=15.0, margin=0.05, count=None):
    """Particles simulated per frame: Poisson around the model's mean (legacy RandomState: one draw per frame, by the
    host), or `count` for every frame (benchmarks with fixed drop counts)."""
    out = np.zeros(n_frames, np.int64)
    for k in range(n_frames):
        if count is not None:
            out[k] = int(count)
            continue
        cam, rate = _frame_settings(options, fallrate, k, min_px, z_far, margin)
        mean = expected_count(cam, rate, min_px, z_far, margin)[0]
        out[k] = int(np.random.RandomState((int(seed) * 1000003 + k) % (2 ** 32)).poisson(mean))
    return out


def make_particles(cam, dgrid, cdf, n, frame, seed, wind_sigma=1.0, margin=0.05, min_px=1.0, z_far=15.0):
    """The n particles of simulated frame `frame` as PARTICLE_DTYPE records: the numpy statement of
    rr_particles.h make_particle (same operations, same order)."""
    rec = np.zeros(n, PARTICLE_DTYPE)
    if n == 0:
        return rec
    i = np.arange(n, dtype=np.uint64)
    k0, k1 = _key(seed)
    a = philox4x32(i, frame, 0, 0, k0, k1)
    b = philox4x32(i, frame, 1, 0, k0, k1)
    c = philox4x32(i, frame, 2, 0, k0, k1)
    W, H = float(cam.W), float(cam.H)
    D = sample_diameter(dgrid, cdf, unit32(a[0]))                                # mm
    wd = D * 1e-3
    z_max = np.minimum((wd * cam.fpx) / min_px, z_far)
    u1, u2, u3 = unit32(a[1]), unit32(a[2]), unit32(a[3])
    depth = np.maximum(z_max * np.maximum(np.maximum(u1, u2), u3), 0.05)        # uniform in the frustum volume
    lo_x, hi_x, lo_y, hi_y = -margin * W, (1.0 + margin) * W, -margin * H, (1.0 + margin) * H
    px = lo_x + (hi_x - lo_x) * unit32(b[0])
    py = lo_y + (hi_y - lo_y) * unit32(b[1])                                     # from the bottom
    X = ((px - W / 2.0) * depth) / cam.fpx
    Y = ((py - H / 2.0) * depth) / cam.fpx
    Z = -depth
    s4 = ((unit32(c[0]) + unit32(c[1])) + (unit32(c[2]) + unit32(c[3]))) - 2.0
    wind = (s4 * 1.7320508075688772) * wind_sigma
    t = cam.exposure
    X2 = X + wind * t
    Y2 = Y - terminal_velocity(D) * t
    Z2 = Z + cam.speed * t
    depth2 = np.maximum(-Z2, 0.05)
    rec['pid'] = np.arange(n)
    rec['wp1'] = np.stack([X, Y, Z], axis=1)
    rec['wp2'] = np.stack([X2, Y2, Z2], axis=1)
    rec['wd1'] = rec['wd2'] = wd
    rec['ip1'] = np.stack([px, py], axis=1)
    rec['ip2'] = np.stack([W / 2.0 + (cam.fpx * X2) / depth2, H / 2.0 + (cam.fpx * Y2) / depth2], axis=1)
    rec['iw1'] = (wd * cam.fpx) / depth
    rec['iw2'] = (wd * cam.fpx) / depth2
    return rec


def generate(options, fallrate, n_frames, seed=0, min_px=1.0, z_far=15.0, margin=0.05, wind_sigma=1.0, count=None):
    """(frames, drops) record arrays of `n_frames` camera frames.  `count`: force that many drops per frame instead
    of the Poisson-distributed physical count."""
    frames = np.zeros(n_frames, PARTICLE_FRAME_DTYPE)
    counts = frame_counts(options, fallrate, n_frames, seed, min_px, z_far, margin, count)
    chunks = []
    first = 0
    tables = {}
    for k in range(n_frames):
        cam, rate = _frame_settings(options, fallrate, k, min_px, z_far, margin)
        tk = (rate, cam.fpx, cam.W, cam.H)
        if tk not in tables:
            tables[tk] = expected_count(cam, rate, min_px, z_far, margin)
        _, dgrid, cdf, _ = tables[tk]
        n = int(counts[k])
        rec = make_particles(cam, dgrid, cdf, n, k, seed, wind_sigma, margin, min_px, z_far)
        frames[k] = (k, int(round(cam.exposure * 1e6)), int(round(k * 1e6 / cam.hz)), n, first, n)
        chunks.append(rec)
        first += n
    return frames, (np.concatenate(chunks) if chunks else np.zeros(0, PARTICLE_DTYPE))


# ---- the same run described to the library (rr_set_particle_tables / rr_sim_frame) -----------------------------------
def diameter_tables(options, fallrate, n_frames, min_px=1.0, z_far=15.0, margin=0.05):
    """(d_grid [N_GRID], cdf [n_tables, N_GRID], table index per frame): one table per distinct (fall rate, camera)."""
    keys, tabs, idx = {}, [], np.zeros(n_frames, np.int32)
    dgrid = None
    for k in range(n_frames):
        cam, rate = _frame_settings(options, fallrate, k, min_px, z_far, margin)
        tk = (rate, cam.fpx, cam.W, cam.H)
        if tk not in keys:
            _, dgrid, cdf, _ = expected_count(cam, rate, min_px, z_far, margin)
            keys[tk] = len(tabs)
            tabs.append(cdf)
        idx[k] = keys[tk]
    return dgrid, np.ascontiguousarray(np.stack(tabs)), idx


def sim_frames(options, fallrate, n_frames, render_scale=1, seed=0, draw_seeds=None, min_px=1.0, z_far=15.0, margin=0.05,
               wind_sigma=1.0, count=None, frame_ids=None):
    """SIM_FRAME_DTYPE records (hip_backend: the numpy mirror of rr_sim_frame) of `n_frames` camera frames + the tables
    they refer to: (sims, d_grid, cdf).  draw_seeds: np.random.seed(...) of the renderer's per-drop draws per frame
    (generator.py:318: the frame's index; default: the frame number)."""
    from .. import hip_backend
    dgrid, cdf, tab = diameter_tables(options, fallrate, n_frames, min_px, z_far, margin)
    counts = frame_counts(options, fallrate, n_frames, seed, min_px, z_far, margin, count)
    sims = np.zeros(n_frames, hip_backend.SIM_FRAME_DTYPE)
    k0, k1 = _key(seed)
    for k in range(n_frames):
        cam, _ = _frame_settings(options, fallrate, k, min_px, z_far, margin)
        s = sims[k]
        s['sensor_w'], s['sensor_h'], s['render_scale'], s['n_particles'] = cam.W, cam.H, render_scale, counts[k]
        s['key0'], s['key1'], s['frame'] = k0, k1, k if frame_ids is None else frame_ids[k]
        s['draw_seed'] = k if draw_seeds is None else draw_seeds[k]
        s['table'] = tab[k]
        s['fpx'], s['exposure_s'], s['speed_mps'] = cam.fpx, cam.exposure, cam.speed
        s['wind_sigma'], s['margin'], s['min_px'], s['z_far'] = wind_sigma, margin, min_px, z_far
    return sims, dgrid, cdf


def expected_records(sims, dgrid, cdf, db, dataset='kitti'):
    """What rr_generate_drops_device must leave in HBM for these frames: per frame the rr_drop records (DROP_DTYPE) made the
    host's way -- make_particles -> DBManager.load_streaks_from_records (the loader's derived fields) ->
    hip_backend.pack_frame (frame filter + the frame's random draws) with the exact rotation terms.  `db`: a DBManager
    with the streak database loaded (texture ratios)."""
    from .. import hip_backend
    from ..common import bad_weather as bw
    out = []
    for s in sims:
        cam = type('Cam', (), dict(W=int(s['sensor_w']), H=int(s['sensor_h']), fpx=float(s['fpx']), exposure=float(s['exposure_s']),
                                   speed=float(s['speed_mps'])))()
        seed = int(s['key0']) | (int(s['key1']) << 32)
        rec = make_particles(cam, dgrid, cdf[int(s['table'])], int(s['n_particles']), int(s['frame']), seed, float(s['wind_sigma']),
                             float(s['margin']), float(s['min_px']), float(s['z_far']))
        fr = np.zeros(1, PARTICLE_FRAME_DTYPE)
        fr[0] = (0, 0, 0, len(rec), 0, len(rec))
        m = bw.DBManager()
        m.ratio = db.ratio
        rs = int(s['render_scale'])
        W, H = int(s['sensor_w']) // rs, int(s['sensor_h']) // rs
        m.load_streaks_from_records(fr, rec, dataset, {"render_scale": rs}, [W, H])
        out.append(hip_backend.pack_frame(m.streaks_simulator[0].table, m, W, H, int(s['draw_seed']), rotation='exact'))
    return out


def write_xml(path, frames, drops):
    """The file the reference's DBManager.load_streaks_from_xml reads (bad_weather.py:192-211).  Written under a temporary
    name and moved into place: a reader never sees half a file."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    fmt = ('    <streak pid="%d" wp1="(%.17g;%.17g;%.17g)" wp2="(%.17g;%.17g;%.17g)" wd1="%.17g" wd2="%.17g" '
           'ip1="(%.17g;%.17g)" ip2="(%.17g;%.17g)" iw1="%.17g" iw2="%.17g"/>')
    tmp = '%s.tmp%d' % (path, os.getpid())
    with open(tmp, 'w') as fh:
        fh.write('<?xml version="1.0" ?>\n<simulation>\n')
        for fr in frames:
            a, n = int(fr['first_drop']), int(fr['n_drops'])
            fh.write('  <frame id="%d" t="%d" d="%d" rs="%d">\n' % (fr['id'], fr['t'], fr['d'], fr['rs']))
            d = drops[a:a + n]
            if n:
                cols = np.column_stack([d['pid'].astype(np.float64), d['wp1'], d['wp2'], d['wd1'], d['wd2'], d['ip1'], d['ip2'],
                                        d['iw1'], d['iw2']])
                fh.write('\n'.join(fmt % ((int(r[0]),) + tuple(r[1:])) for r in cols.tolist()))
                fh.write('\n')
            fh.write('  </frame>\n')
        fh.write('</simulation>\n')
    os.replace(tmp, path)
    return path


def n_sim_frames(options):
    steps = options.get("sim_steps", {}) or {}
    n_steps = max([len(v) for v in steps.values()] + [0])
    return n_steps if options.get("sim_mode") == "steps" and n_steps else int(options["sim_duration"] * options["cam_hz"])


def simulate(sim, weather, n_frames=None, seed=0, force_recompute=False):
    """The role of the reference's tools/particles_simulation.process for ONE sequence: `sim` = common.db.sim(...)
    ({"path", "options"}), `weather` = {"weather": "rain", "fallrate": R}.  Writes
    <sim path>/<weather>/<R>mm/sim_camera0.xml unless it exists; returns its path."""
    options = sim["options"]
    out_dir = os.path.join(sim["path"], weather["weather"], '{}mm'.format(weather["fallrate"]))
    path = os.path.join(out_dir, 'sim_camera0.xml')
    if os.path.exists(path) and not force_recompute:
        return path
    if n_frames is None:
        n_frames = n_sim_frames(options)
    frames, drops = generate(options, weather["fallrate"], n_frames, seed=seed)
    return write_xml(path, frames, drops)
