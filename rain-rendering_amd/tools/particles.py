"""Rain-particle generator: the stand-in for the reference's external simulator.

The reference drives a closed-source binary (``3rdparty/weather-particle-simulator/.../AHLSimulation``) through
``tools/simulation.py`` / ``tools/particles_simulation.py`` with the settings of ``common/db.py:41-70``
(``cam_hz``, ``cam_CCD_WH``, ``cam_CCD_pixsize``, ``cam_focal``, ``cam_exposure``, ``sim_mode``, ``sim_steps``
{``cam_motion``, ``cam_exposure``, ``cam_focal``, ``rain_fallrate``}, ``sim_duration``) and reads its output back as
``<particles>/<dataset>/<sequence>/rain/<N>mm/*_camera0.xml`` (schema: bad_weather.py:192-211).  There is no source
for that binary, so there is no oracle for this module: it is a physically motivated generator, validated
statistically (tests/test_particles.py), that takes the SAME settings and emits the SAME schema -- either as the XML
file the reference's loader reads, or directly as the record arrays ``DBManager.load_streaks_from_records`` takes
(BASELINE config 5: "no XML").

Model (per camera frame, camera at the origin looking along -z, x right, y up, image origin bottom-left -- the
conventions the loader undoes, bad_weather.py:221-224):
  * drop diameters follow Marshall & Palmer (1948): N(D) = N0 exp(-Lambda D), N0 = 8000 m^-3 mm^-1,
    Lambda = 4.1 R^-0.21 mm^-1 for a fall rate R in mm/hr, D in [0.5, 6] mm;
  * only drops that can appear at least `min_px` wide are simulated: depth z <= D f / (pixel * min_px) (and <= z_far),
    uniformly in the viewing frustum up to that depth -- the expected count is the integral of N(D) over that volume,
    the actual count of a frame is Poisson distributed;
  * a drop falls at its terminal velocity v(D) = 9.65 - 10.3 exp(-0.6 D) m/s (Atlas et al. 1973), drifts with a
    horizontal wind and approaches the camera at the vehicle's speed (``sim_steps['cam_motion']``, km/h);
    the streak is the path covered during the exposure, both ends projected through the pinhole camera;
  * image widths are D f / (pixel z) at either end.
"""
import os

import numpy as np

from ..common.bad_weather import PARTICLE_DTYPE, PARTICLE_FRAME_DTYPE

N0 = 8000.0                      # m^-3 mm^-1
D_MIN, D_MAX = 0.5, 6.0          # mm


def mp_lambda(fallrate):
    """Marshall-Palmer slope, mm^-1."""
    return 4.1 * float(fallrate) ** -0.21


def terminal_velocity(d_mm):
    """m/s (Atlas, Srivastava & Sekhon 1973)."""
    return 9.65 - 10.3 * np.exp(-0.6 * np.asarray(d_mm, np.float64))


class FrameCamera:
    def __init__(self, options, step):
        """The camera of simulation step `step`: `options` is what common.db.sim() returns under "options"."""
        steps = options.get("sim_steps", {}) or {}

        def stepped(key, default):
            v = steps.get(key)
            if v is None or len(v) == 0:
                return default
            return float(v[min(step, len(v) - 1)])           # a parameter stays applied unless later changed (db.py)
        self.W, self.H = options["cam_CCD_WH"]
        self.pix = options["cam_CCD_pixsize"] * 1e-6
        self.focal = stepped("cam_focal", options["cam_focal"]) * 1e-3
        self.exposure = stepped("cam_exposure", options["cam_exposure"]) * 1e-3
        self.speed = stepped("cam_motion", 0.0) / 3.6         # km/h -> m/s
        self.fpx = self.focal / self.pix
        self.hz = options["cam_hz"]


def expected_count(cam, fallrate, min_px=1.0, z_far=15.0, margin=0.05, n_grid=512):
    """(expected visible drops per frame, diameter grid, its sampling CDF, z_max per diameter)."""
    lam = mp_lambda(fallrate)
    d = np.linspace(D_MIN, D_MAX, n_grid)
    z_max = np.minimum(d * 1e-3 * cam.fpx / min_px, z_far)
    area = (1 + 2 * margin) ** 2 * cam.W * cam.H / cam.fpx ** 2          # frustum cross-section at unit depth
    dens = N0 * np.exp(-lam * d) * area * z_max ** 3 / 3.0               # drops per mm of diameter
    cdf = np.concatenate([[0.0], np.cumsum(0.5 * (dens[1:] + dens[:-1]) * np.diff(d))])
    return float(cdf[-1]), d, cdf / cdf[-1], z_max


def generate(options, fallrate, n_frames, seed=0, min_px=1.0, z_far=15.0, margin=0.05, wind_sigma=1.0, count=None):
    """(frames, drops) record arrays of `n_frames` camera frames.  `count`: force that many drops per frame instead
    of the Poisson-distributed physical count (benchmarks with fixed drop counts)."""
    frames = np.zeros(n_frames, PARTICLE_FRAME_DTYPE)
    chunks = []
    first = 0
    steps = options.get("sim_steps", {}) or {}
    for k in range(n_frames):
        cam = FrameCamera(options, k)
        rate = float(steps["rain_fallrate"][min(k, len(steps["rain_fallrate"]) - 1)]) if len(steps.get("rain_fallrate", ())) else fallrate
        rng = np.random.RandomState((int(seed) * 1000003 + k) % (2 ** 32))
        mean, dgrid, cdf, zmax_grid = expected_count(cam, rate, min_px, z_far, margin)
        n = int(count) if count is not None else int(rng.poisson(mean))
        rec = np.zeros(n, PARTICLE_DTYPE)
        if n:
            D = np.interp(rng.rand(n), cdf, dgrid)                               # mm
            z_max = np.minimum(D * 1e-3 * cam.fpx / min_px, z_far)
            depth = np.maximum(z_max * rng.rand(n) ** (1.0 / 3.0), 0.05)         # uniform in the frustum volume
            px = rng.uniform(-margin * cam.W, (1 + margin) * cam.W, n)
            py = rng.uniform(-margin * cam.H, (1 + margin) * cam.H, n)          # from the bottom
            X = (px - cam.W / 2) * depth / cam.fpx
            Y = (py - cam.H / 2) * depth / cam.fpx
            Z = -depth
            t = cam.exposure
            X2 = X + rng.normal(0.0, wind_sigma, n) * t
            Y2 = Y - terminal_velocity(D) * t
            Z2 = Z + cam.speed * t
            depth2 = np.maximum(-Z2, 0.05)
            rec['pid'] = np.arange(n)
            rec['wp1'] = np.stack([X, Y, Z], axis=1)
            rec['wp2'] = np.stack([X2, Y2, Z2], axis=1)
            rec['wd1'] = rec['wd2'] = D * 1e-3
            rec['ip1'] = np.stack([px, py], axis=1)
            rec['ip2'] = np.stack([cam.W / 2 + cam.fpx * X2 / depth2, cam.H / 2 + cam.fpx * Y2 / depth2], axis=1)
            rec['iw1'] = D * 1e-3 * cam.fpx / depth
            rec['iw2'] = D * 1e-3 * cam.fpx / depth2
        frames[k] = (k, int(round(cam.exposure * 1e6)), int(round(k * 1e6 / cam.hz)), n, first, n)
        chunks.append(rec)
        first += n
    return frames, (np.concatenate(chunks) if chunks else np.zeros(0, PARTICLE_DTYPE))


def write_xml(path, frames, drops):
    """The file the reference's DBManager.load_streaks_from_xml reads (bad_weather.py:192-211)."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    fmt = ('    <streak pid="%d" wp1="(%.17g;%.17g;%.17g)" wp2="(%.17g;%.17g;%.17g)" wd1="%.17g" wd2="%.17g" '
           'ip1="(%.17g;%.17g)" ip2="(%.17g;%.17g)" iw1="%.17g" iw2="%.17g"/>')
    with open(path, 'w') as fh:
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
    return path


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
        steps = options.get("sim_steps", {}) or {}
        n_steps = max([len(v) for v in steps.values()] + [0])
        n_frames = n_steps if options.get("sim_mode") == "steps" and n_steps else int(options["sim_duration"] * options["cam_hz"])
    frames, drops = generate(options, weather["fallrate"], n_frames, seed=seed)
    return write_xml(path, frames, drops)
