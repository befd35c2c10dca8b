"""Host-side mirror of the reference's common/bad_weather.py for the hot path: the input
types (Streak, Frame, DropType), the two loaders of DBManager and its RNG-driven texture
pick.  Same names, argument meaning and error behaviour as the reference; the per-drop
arithmetic itself lives in the HIP library (csrc/), not here.

In addition to the reference's object model (dict pid -> Streak) every Frame carries a
packed, column-wise ``StreakTable`` so that filtering and packing 10^4 drops per frame
does not walk Python objects.  Streak attributes are numpy *views* into that table, so
the reference's in-place endpoint mutation (generator.py:152-161) stays visible to both.
"""
import os
import sys
from enum import Enum
from xml.etree.ElementTree import parse

import numpy as np

from . import my_utils


class DropType(Enum):          # reference bad_weather.py:40-43
    Big = 0
    Medium = 1
    Small = 2


class Streak:                  # reference bad_weather.py:46-63
    def __init__(self):
        self.pid = None
        self.world_position_start = None
        self.world_position_end = None
        self.world_diameter_start = None
        self.world_diameter_end = None
        self.image_position_start = None
        self.image_position_end = None
        self.image_diameter_start = None
        self.image_diameter_end = None
        self.ratio = None
        self.max_width = None
        self.length = None
        self.drop_type = None

    def __repr__(self):
        return str(self.__dict__).replace(',', '\n')


class StreakTable:
    """Column store of the streaks of one simulated frame (kept streaks only, file order,
    later duplicates of a pid overwrite earlier ones exactly like dict.update does)."""

    FIELDS = ('pid', 'wps', 'wpe', 'wd1', 'wd2', 'ips', 'ipe', 'iw1', 'iw2', 'ratio', 'max_width', 'length', 'type')

    def __init__(self, n):
        self.pid = np.zeros(n, np.int64)
        self.wps = np.zeros((n, 3))
        self.wpe = np.zeros((n, 3))
        self.wd1 = np.zeros(n)
        self.wd2 = np.zeros(n)
        self.ips = np.zeros((n, 2), np.int64)
        self.ipe = np.zeros((n, 2), np.int64)
        self.iw1 = np.zeros(n)
        self.iw2 = np.zeros(n)
        self.ratio = np.zeros(n)
        self.max_width = np.zeros(n, np.int64)
        self.length = np.zeros(n, np.int64)
        self.type = np.zeros(n, np.int32)

    def __len__(self):
        return len(self.pid)

    def take(self, idx):
        """A new table (own storage, also for slices) of the rows idx."""
        t = StreakTable(0)
        for k in self.FIELDS:
            setattr(t, k, np.array(getattr(self, k)[idx], copy=True))
        return t

    def streak(self, i):
        s = Streak()
        s.pid = int(self.pid[i])
        s.world_position_start = self.wps[i]
        s.world_position_end = self.wpe[i]
        s.world_diameter_start = float(self.wd1[i])
        s.world_diameter_end = float(self.wd2[i])
        s.image_position_start = self.ips[i]
        s.image_position_end = self.ipe[i]
        s.image_diameter_start = float(self.iw1[i])
        s.image_diameter_end = float(self.iw2[i])
        s.ratio = float(self.ratio[i])
        s.max_width = int(self.max_width[i])
        s.length = self.length[i]
        s.drop_type = DropType(int(self.type[i]))
        return s


class Frame:                   # reference bad_weather.py:66-75
    def __init__(self):
        self.id = None
        self.starting_time = None
        self.exposure_time = None
        self.streaks_count = None
        self._streaks = None
        self.table = None

    @property
    def streaks(self):
        """dict pid -> Streak (views into self.table), built on first use."""
        if self._streaks is None:
            self._streaks = {int(self.table.pid[i]): self.table.streak(i) for i in range(len(self.table))}
        return self._streaks

    def __repr__(self):
        return str({'id': self.id, 'n': len(self.table)})


def _parse_vec(s):
    return [float(v) for v in s[1:-1].split(';')]


_VEC_SEPARATORS = str.maketrans('()[];,', '      ')


def _bulk_vec(strings, k):
    """["(a;b;c)", ...] or ["a", ...] -> float64 (n, k)."""
    if not strings:
        return np.zeros((0, k))
    vals = np.array(' '.join(strings).translate(_VEC_SEPARATORS).split(), dtype=np.float64)
    if vals.size != len(strings) * k:
        raise ValueError("malformed vector attribute in the particles file")
    return vals.reshape(len(strings), k)


PARTICLE_FRAME_DTYPE = np.dtype([('id', '<i8'), ('t', '<i8'), ('d', '<i8'), ('rs', '<i8'), ('first_drop', '<i8'), ('n_drops', '<i8')])
PARTICLE_DTYPE = np.dtype([('pid', '<i8'), ('wp1', '<f8', (3,)), ('wp2', '<f8', (3,)), ('wd1', '<f8'), ('wd2', '<f8'),
                           ('ip1', '<f8', (2,)), ('ip2', '<f8', (2,)), ('iw1', '<f8'), ('iw2', '<f8')])


def _read_particles_native(path):
    """(frames, drops) record arrays from librainhip's rr_host_parse_particles; None when the file uses XML the
    native parser leaves to a full parser (RR_E_UNSUPPORTED); raises on a malformed file."""
    import ctypes
    from .. import hip_backend
    lib = hip_backend.load_library()
    assert lib.rr_sizeof_particle() == PARTICLE_DTYPE.itemsize and lib.rr_sizeof_particle_frame() == PARTICLE_FRAME_DTYPE.itemsize
    nf, nd = ctypes.c_int64(0), ctypes.c_int64(0)
    cap_f, cap_d = 0, 0
    frames, drops = np.zeros(0, PARTICLE_FRAME_DTYPE), np.zeros(0, PARTICLE_DTYPE)
    for _ in range(2):                       # count, then fill
        rc = lib.rr_host_parse_particles(os.fsencode(path), frames.ctypes.data_as(ctypes.c_void_p), cap_f,
                                         drops.ctypes.data_as(ctypes.c_void_p), cap_d, ctypes.byref(nf), ctypes.byref(nd))
        if rc == -7:                         # RR_E_UNSUPPORTED
            return None
        if rc != 0:
            raise ValueError("rr_host_parse_particles(%s) failed with %d" % (path, rc))
        if nf.value <= cap_f and nd.value <= cap_d:
            return frames[:nf.value], drops[:nd.value]
        cap_f, cap_d = nf.value, nd.value
        frames, drops = np.zeros(cap_f, PARTICLE_FRAME_DTYPE), np.zeros(cap_d, PARTICLE_DTYPE)
    raise RuntimeError("particles file changed while it was being read")


def _read_particles_etree(path):
    """The same records through xml.etree (files the native parser does not take)."""
    simulation = parse(path).getroot()
    fr_rows, chunks, n_tot = [], [], 0
    for frame in simulation:
        att = [d.attrib for d in frame]
        n = len(att)
        rec = np.zeros(n, PARTICLE_DTYPE)
        if n:
            rec['pid'] = [int(a["pid"]) for a in att]
            rec['wp1'] = _bulk_vec([a["wp1"] for a in att], 3)
            rec['wp2'] = _bulk_vec([a["wp2"] for a in att], 3)
            rec['wd1'] = _bulk_vec([a['wd1'] for a in att], 1)[:, 0]
            rec['wd2'] = _bulk_vec([a['wd2'] for a in att], 1)[:, 0]
            rec['ip1'] = _bulk_vec([a["ip1"] for a in att], 2)
            rec['ip2'] = _bulk_vec([a["ip2"] for a in att], 2)
            rec['iw1'] = _bulk_vec([a['iw1'] for a in att], 1)[:, 0]
            rec['iw2'] = _bulk_vec([a['iw2'] for a in att], 1)[:, 0]
        fr_rows.append((int(frame.attrib['id']), int(frame.attrib['t']), int(frame.attrib['d']), int(frame.attrib['rs']), n_tot, n))
        chunks.append(rec)
        n_tot += n
    frames = np.array(fr_rows, PARTICLE_FRAME_DTYPE) if fr_rows else np.zeros(0, PARTICLE_FRAME_DTYPE)
    return frames, (np.concatenate(chunks) if chunks else np.zeros(0, PARTICLE_DTYPE))


def _read_particles(path):
    out = _read_particles_native(path)
    return out if out is not None else _read_particles_etree(path)


class DBManager:
    def __init__(self, streaks_path=None, streaks_path_xml=None, norm_coeff_path=None):
        """Same constructor as the reference (bad_weather.py:79-91)."""
        self.streaks_path = streaks_path
        self.streaks_path_xml = streaks_path_xml
        self.streaks_light = []
        self.norm_coeff_path = norm_coeff_path
        self.streaks_simulator = {}
        self.ratio = np.array([])

    @staticmethod
    def classify_drop(w):      # reference bad_weather.py:99-106
        if w >= 4:
            return DropType(0)
        if w > 1:
            return DropType(1)
        return DropType(2)

    def load_streak_database(self):
        """reference bad_weather.py:108-146.  Textures are kept as a list of HxW uint8 gray
        arrays (the reference replicates them to 3 identical channels and builds a ragged
        np.array, which modern numpy refuses)."""
        from PIL import Image
        if not os.path.exists(self.streaks_path):
            print("No existing path for streak database (", self.streaks_path, ")")
            sys.exit(-1)
        norm_coeffs = {}
        with open(self.norm_coeff_path, 'r') as fh:
            lines = fh.readlines()
        coeff = None
        for line in lines:
            if line[:2] == 'cv':
                coeff = int(line[2:])
                continue
            norm_coeffs.update({coeff: [float(v) for v in line.split('\n')[0].split(' ')[:-1]]})
        tmp = []
        ratio = []
        for file_name in my_utils.os_listdir(self.streaks_path):
            name = os.path.splitext(file_name)[0]
            coeff, osc = name.split('_')
            coeff = int(coeff[-1:]) if len(coeff) == 3 else int(coeff[-2:])
            osc = int(osc[-1:])
            img = np.array(Image.open(os.path.join(self.streaks_path, file_name)))      # 16-bit gray
            tex = ((255.0 * norm_coeffs[coeff][osc] * img) / 65535.0).astype(np.uint8)
            tmp.append(np.ascontiguousarray(tex))
            ratio.append(tex.shape[1] / tex.shape[0])
        self.ratio = np.unique(np.array(ratio))
        self.streaks_light = tmp

    def load_streaks_from_xml(self, dataset, settings, image_shape_WH, use_pickle=True, verbose=True):
        """reference bad_weather.py:148-248 (the pickle cache is read-only in the reference and
        its call site passes use_pickle=False, generator.py:281; not implemented).

        The file is read by the library's native parser (rr_host_parse_particles: raw attribute values, strtod /
        strtoll -- the doubles Python's float() gives); an XML construct outside the simulator's subset (DOCTYPE,
        CDATA, entity references) goes through xml.etree instead.  Everything derived follows in bulk below."""
        print('Reading particles file {}'.format(self.streaks_path_xml))
        if not os.path.exists(self.streaks_path_xml):
            print("No existing path for XML file (" + str(self.streaks_path_xml) + ")")
            sys.exit(-1)
        try:
            frames, drops = _read_particles(self.streaks_path_xml)
        except Exception:
            raise Exception("Reading XML file {} crashed, which is likely due to corrupted particles simulation "
                            "files. If so, delete this simulation folder manually and re-run to allow generation "
                            "of new simulation.".format(self.streaks_path_xml))
        self.load_streaks_from_records(frames, drops, dataset, settings, image_shape_WH)

    def load_streaks_from_records(self, frames, drops, dataset, settings, image_shape_WH):
        """The derived fields of reference bad_weather.py:208-241 for raw particle records (PARTICLE_FRAME_DTYPE /
        PARTICLE_DTYPE arrays: what rr_host_parse_particles reads from an XML file, or what tools/particles.py
        generates without one), every drop of the sequence at once."""
        rs = settings["render_scale"]
        gan = dataset == 'nuscenes_gan'
        r_gan = np.mean((image_shape_WH[0] / 1600, image_shape_WH[1] / 900)) if gan else None
        try:
            pid = drops['pid']
            wps, wpe = drops['wp1'].copy(), drops['wp2'].copy()
            iw = np.stack([drops['iw1'], drops['iw2']], axis=1)
            if gan:
                ips, ipe, iws = drops['ip1'] * r_gan, drops['ip2'] * r_gan, iw * r_gan
            else:
                ips, ipe, iws = drops['ip1'] / rs, drops['ip2'] / rs, iw / rs
            ips[:, 1] = image_shape_WH[1] - ips[:, 1]
            ipe[:, 1] = image_shape_WH[1] - ipe[:, 1]
            wps[:, 2] *= -1
            wpe[:, 2] *= -1
            diff = np.abs(ips - ipe)
            max_width = np.maximum(iws[:, 0], iws[:, 1]).astype(np.int64)     # int(max(..)) truncation
            with np.errstate(all='ignore'):
                nrm = np.sqrt(diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1])
                dir2y = -(diff[:, 1] / nrm)
                cos_theta = (diff[:, 0] / nrm) * 0 + dir2y * -1
                actual_length = diff[:, 1] / cos_theta
                ratio = max_width / actual_length
            ipe_i = np.round(ipe).astype(np.int64)
            ips_i = np.round(ips).astype(np.int64)
            dd = (ips_i - ipe_i).astype(np.float64)
            length = np.ceil(np.sqrt(dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1])).astype(np.int64)
            dtype = np.where(max_width >= 4, 0, np.where(max_width > 1, 1, 2)).astype(np.int32)
            keep = (max_width >= 1) & (length >= 1)
            for fr in frames:
                f = Frame()
                f.id, f.exposure_time, f.starting_time, f.streaks_count = int(fr['id']), int(fr['t']), int(fr['d']), int(fr['rs'])
                a, n = int(fr['first_drop']), int(fr['n_drops'])
                kept = a + np.nonzero(keep[a:a + n])[0]
                p_f = pid[kept]
                if len(np.unique(p_f)) != len(p_f):
                    # dict.update semantics: a repeated pid keeps its FIRST position, LAST value
                    order = {}
                    for k in kept:
                        order[int(pid[k])] = k
                    kept = np.fromiter(order.values(), dtype=np.int64, count=len(order))
                t = StreakTable(len(kept))
                t.pid[:] = pid[kept]
                t.wps[:] = wps[kept]
                t.wpe[:] = wpe[kept]
                t.wd1[:] = drops['wd1'][kept]
                t.wd2[:] = drops['wd2'][kept]
                t.ips[:] = ips_i[kept]
                t.ipe[:] = ipe_i[kept]
                t.iw1[:] = iws[kept, 0]
                t.iw2[:] = iws[kept, 1]
                t.ratio[:] = ratio[kept]
                t.max_width[:] = max_width[kept]
                t.length[:] = length[kept]
                t.type[:] = dtype[kept]
                f.table = t
                self.streaks_simulator.update({f.id: f})
        except Exception:
            import traceback
            print('\n[ERROR] Error while parsing XML file.\n\tFile: ' + str(self.streaks_path_xml))
            traceback.print_exc()
            sys.exit(-1)

    def texture_bucket(self, ratio):
        """Which block of ten textures take_drop_texture draws from (bad_weather.py:250-265);
        vectorised: bucket b is the number of DB ratios[0..3] that are <= the drop's ratio,
        with NaN falling through to the last block like the reference's else branch."""
        r = np.asarray(ratio, np.float64)
        b = np.full(r.shape, 4, np.int64)
        for k in (3, 2, 1, 0):
            b = np.where(r < self.ratio[k], k, b)
        return b

    def take_drop_texture_index(self, drop):
        """The RNG draw of take_drop_texture (always exactly one randint per drop)."""
        b = int(self.texture_bucket(drop.ratio))
        return np.random.randint(10 * b, 10 * b + 10)

    def take_drop_texture(self, drop):
        """reference bad_weather.py:250-265: HxWx3 float64 texture in [0,1]."""
        tex = self.streaks_light[self.take_drop_texture_index(drop)] / 255.0
        return np.dstack([tex, tex, tex])


class RainRenderer:
    """Constants holder with the reference's constructor (bad_weather.py:272-278); the
    rendering itself is rr_render_frames in the HIP library."""

    def __init__(self, focal, f_number, focus_plane, radius, fov):
        self.f = focal
        self.N = f_number
        self.focus_plane = focus_plane
        self.radius = radius
        self.fov = fov

    def compute_circle(self, o, is_infinity=False):          # reference bad_weather.py:464-469
        if is_infinity:
            return self.f ** 2 / (self.N * o)
        result = ((o - self.focus_plane) * self.f ** 2) / (o * (self.focus_plane - self.f) * self.N)
        return result / 4.65e-06

    @staticmethod
    def warping_points(drop, drop_texture, image_width, image_height):
        """reference bad_weather.py:300-329: source / destination quads of the Big-drop warp, max and min corner."""
        x0 = round(drop.image_position_start[0])
        x1 = round(drop.image_position_end[0])
        y0 = round(drop.image_position_start[1])
        y1 = round(drop.image_position_end[1])
        d0 = np.floor(drop.image_diameter_start)
        d1 = np.floor(drop.image_diameter_end)
        minx = max(min(x0, x1), 0)
        miny = max(min(y0, y1), 0)
        maxx = min(max(x0 + d0, x1 + d1), image_width)
        maxy = min(max(y0, y1), image_height)
        epsilon = 0.001                                   # to prevent singularity of the perspective matrix
        p1 = np.float32([[0, 0], [drop_texture.shape[1], 0], [drop_texture.shape[1], drop_texture.shape[0]],
                         [0, drop_texture.shape[0]]])
        p2 = np.float32([[x0 - minx, y0 - miny], [x0 - minx + d0, y0 - miny], [x1 - minx + d1 + epsilon, y1 - miny],
                         [x1 - minx + epsilon, y1 - miny]])
        return p1, p2, np.array([maxx, maxy]), np.array([minx, miny])

    def placed_tile(self, drop_minC, tw, th, drop_distance, imW, imH):
        """Where a tw x th tile anchored at drop_minC ends up, as the reference leaves it: (drop_minC, rows, columns) after
        the defocus pad shift = int(10 c) (bad_weather.py:291-295), the clamp of the origin into the frame (:418-419) and
        the crop of what that cut off (:420-422).  drop_distance None = rendering_strategy 'white' (:349-353): no pad,
        no clamp."""
        if drop_distance is None:
            return np.array(drop_minC), th, tw
        shift = int(10 * abs(self.compute_circle(abs(drop_distance))))
        tmp = np.array([int(drop_minC[0]) - shift, int(drop_minC[1]) - shift])
        min_c = np.array([np.clip(tmp[0], 0, imW), np.clip(tmp[1], 0, imH)])
        delta = min_c - tmp                                   # > 0: rows / columns cut off at the top / left border
        ph, pw = th + 2 * shift, tw + 2 * shift
        ph = len(range(ph)[:delta[1]]) if delta[1] < 0 else max(ph - delta[1], 0)
        pw = len(range(pw)[:delta[0]]) if delta[0] < 0 else max(pw - delta[0], 0)
        return min_c, ph, pw

    def _ctx(self, dataset):
        """The library context behind the single-drop seam (created on first use; Generator.run shares its own)."""
        from . import db as settings_db
        from .. import hip_backend
        if getattr(self, '_hip', None) is None:
            self._hip = hip_backend.RainHip(int(os.environ.get('LOCAL_RANK', '0')))
            self._hip_cam = None
            self._hip.set_streak_db([np.zeros((2, 2), np.uint8)])          # the tile comes from the caller: no texture is read
        exposure = settings_db.settings(dataset)["cam_exposure"]            # bad_weather.py:344
        key = (self.f, self.N, exposure, self.focus_plane, self.radius, self.fov)
        if self._hip_cam != key:
            self._hip.set_camera(hip_backend.make_camera(self.f, self.N, exposure, self.focus_plane, self.radius, self.fov))
            self._hip_cam = key
        return self._hip

    def add_drop_to_image(self, dataset, env_map_xyY, solid_angle_map, drop_fov_pts, drop_minC, bg, rainy_bg,
                          rainy_mask, rainy_saturation_mask, drop, drop_dict, irrad_type, rendering_strategy,
                          opacity_attenuation=1.0):
        """The reference's inner seam, same signature and the same SIX return values (bad_weather.py:336-462):
        composite ONE caller-made tile `drop` (H x W x 4, gray with alpha, as Generator.compute_drop builds it) whose
        field-of-view polygon is `drop_fov_pts` and whose position is `drop_minC` into rainy_bg / rainy_mask -- in place
        and returned -- and hand back

            (rainy_bg, rainy_mask, rainy_saturation_mask, drop_vis, drop_blend, drop_minC)

        as the reference does (:462): `drop_vis` the coloured, defocused tile cropped to the frame (h x w x 4: alpha =
        what was added to the mask, colour = alpha x the drop's colour constants K -- shape and alpha channel are the
        reference's; its COLOUR channels are blur(K * [raw alpha > 0]) (it paints K where the raw tile is positive and
        blurs all four channels, :378-381,286-298), which differs from K * blur(alpha) wherever the raw tile has fractional
        alpha; the only consumer, make_rain_layer's rain_layer, is a dead output), `drop_blend` the blended image region
        under it (h x w x 3) and `drop_minC` the tile's clamped position (:418-419) -- what the reference's caller passes
        on to make_rain_layer (generator.py:437-438).  Colour from the environment map, defocus, placement, blend and mask
        accumulation run in the library (rr_ext_tile entry of rr_render_frames: one launch chain per call -- use
        Generator.run / rr_render_frames for throughput).

        Like the reference, a drop that cannot be rendered (empty polygon, polygon off the map, non-finite circle of
        confusion) raises; Generator.compute_drop catches that (generator.py:180-189).  rainy_saturation_mask is passed
        through untouched: it is a dead output in the reference (never read after the loop)."""
        from .. import hip_backend
        hip = self._ctx(dataset)
        tile = np.asarray(drop)
        alpha = np.ascontiguousarray(tile[..., 3] if tile.ndim == 3 else tile, np.float64)
        rec = np.zeros(1, hip_backend.DROP_DTYPE)
        rec['x0'], rec['y0'] = int(drop_minC[0]), int(drop_minC[1])
        rec['x1'], rec['y1'] = rec['x0'], rec['y0']
        rec['max_width'] = int(max(drop_dict.image_diameter_start, drop_dict.image_diameter_end))
        rec['length'] = int(drop_dict.length)
        rec['type'] = 0
        rec['iw1'], rec['iw2'] = drop_dict.image_diameter_start, drop_dict.image_diameter_end
        rec['rot_cos'] = 1.0
        white = rendering_strategy == 'white'
        if not white:
            rec['wps'], rec['wpe'] = drop_dict.world_position_start, drop_dict.world_position_end
        elif getattr(drop_dict, 'world_position_start', None) is not None:
            rec['wps'], rec['wpe'] = drop_dict.world_position_start, drop_dict.world_position_end
        poly = None if white else np.asarray(drop_fov_pts, np.float64).reshape(-1, 2)
        out = hip.render_frames([dict(bg=bg, rainy_bg=rainy_bg, env_xyY=env_map_xyY, omega=solid_angle_map, drops=rec,
                                      ext=[dict(alpha=alpha, minC=(int(drop_minC[0]), int(drop_minC[1])), poly=poly)],
                                      opacity_attenuation=opacity_attenuation, strategy=1 if white else 0)],
                                want_colour=True)[0]
        if out['status'][0] != 0:
            raise IndexError("drop not rendered (status %d: 1 no field of view, 2 field of view off the map, 3/4 circle of "
                             "confusion)" % out['status'][0])
        rainy_bg[...] = out['rainy_bg']
        rainy_mask += out['mask']
        min_c, th, tw = self.placed_tile(drop_minC, alpha.shape[1], alpha.shape[0], None if white else drop_dict.world_position_start[2],
                                         np.asarray(bg).shape[1], np.asarray(bg).shape[0])
        y0, x0 = int(min_c[1]), int(min_c[0])
        drop_blend = rainy_bg[y0:y0 + th, x0:x0 + tw, :].copy()
        if white:
            drop_vis = tile[:drop_blend.shape[0], :drop_blend.shape[1]]
        else:
            a_vis = out['mask'][y0:y0 + th, x0:x0 + tw]
            drop_vis = np.dstack([a_vis * out['colour'][0, 0], a_vis * out['colour'][0, 1], a_vis * out['colour'][0, 2], a_vis])
        return rainy_bg, rainy_mask, rainy_saturation_mask, drop_vis, drop_blend, min_c

    @staticmethod
    def make_rain_layer(drop, blended_drop, rain_layer, mask, drop_min_C):
        """reference bad_weather.py:482-495 (a dead output there: rain_layer is never read after the loop; kept so that a
        reference-shaped caller of the seam above runs unchanged): where the mask is wet under the tile, alpha 255 and the
        channel-wise maximum of the layer and the blended drop."""
        x, y = int(drop_min_C[0]), int(drop_min_C[1])
        h, w = drop.shape[:2]
        region = rain_layer[y:y + h, x:x + w]
        wet = mask[y:y + h, x:x + w] > 0
        region[..., 3][wet] = 255
        region[..., :3][wet] = np.maximum(region[..., :3][wet], blended_drop[wet])
        return rain_layer


class FovComputation:
    """reference bad_weather.py:497-704: the polygon on the lat-long environment map seen from a drop.  Host-side
    restatement with the reference's method name and return value (points, [], [], []) for callers of the single-drop
    seam; the batched path evaluates the same arithmetic on the device (csrc/rr_device.h fov_vertex / k_fov_spans)."""

    def __init__(self, camera):
        self.camera = camera

    @staticmethod
    def rotation_matrix(axis, theta):                       # bad_weather.py:532-538 (Rodrigues)
        axis = np.asarray(axis)
        c, s = np.cos(theta), np.sin(theta)
        skv = np.roll(np.roll(np.diag(axis.flatten()), 1, 1), -1, 0)
        return (c * np.identity(3)) + s * (skv - skv.T) + ((1 - c) * np.outer(axis, axis))

    @staticmethod
    def intersection_sphere(position, direction, radius):   # bad_weather.py:540-568, sphere about the origin, far root
        dx, dy, dz = direction
        x0, y0, z0 = position
        a = dx * dx + dy * dy + dz * dz
        b = 2 * dx * x0 + 2 * dy * y0 + 2 * dz * z0
        c = 0 + x0 * x0 + y0 * y0 + z0 * z0 + -2 * 0 - radius * radius
        t1 = (-b + np.sqrt(b ** 2 - 4 * a * c)) / (2 * a)
        return position + (t1 * direction)

    @staticmethod
    def cart2sph(p):                                        # bad_weather.py:570-586
        x, y, z = p
        el = np.arctan2(z, np.sqrt(x ** 2 + y ** 2))
        az = np.arctan2(y, x)
        if az < 0:
            az += 2 * np.pi
        if el < 0:
            el += 2 * np.pi
        if az > np.pi * 2:
            az -= 2 * np.pi
        if el > np.pi * 2:
            el -= 2 * np.pi
        return az, el

    def compute_fov_plane_points(self, drop, radius, fov, N, env_shape):
        """bad_weather.py:596-704.  Returns (points (M, 2) float, [], [], []) with M = N or N + 4, or ([], [], [], [])
        where the reference's `except:` fires.  Same operations in the same order as the reference (pinned bit for bit
        by tests/golden)."""
        try:
            with np.errstate(all='ignore'):
                pos = np.array((np.asarray(drop.world_position_start, np.float64) + np.asarray(drop.world_position_end, np.float64)) / 2)
                pos[1], pos[2] = pos[2], pos[1].copy()                       # y <-> z
                n = (pos - self.camera) / np.linalg.norm(pos - self.camera)
                a, b, c = n[0], n[1], n[2]
                d = np.dot(pos, n)
                if b == 0:
                    b = 0.001
                px_, pz_ = pos[1], 0
                point = np.array([px_, (-a * px_ + d - c * pz_) / b, pz_])
                u = (pos - point) / np.linalg.norm(pos - point)
                if not np.all(~np.isnan(u)):
                    return [], [], [], []
                v = np.dot(n, self.rotation_matrix(np.cross(u, n), -np.deg2rad(fov / 2)))
                azs, pts = [], []
                for angle in np.arange(0, 2 * np.pi, (2 * np.pi) / N):
                    P = self.intersection_sphere(pos, np.dot(v, self.rotation_matrix(n, angle)), radius)
                    azimuth, elevation = self.cart2sph(P)
                    azimuth = ((2 * np.pi - azimuth) - np.pi / 2) % (2 * np.pi)
                    elevation = (elevation + np.pi / 2) % (2 * np.pi)
                    azs.append(azimuth)
                    pts.append([azimuth / (2 * np.pi) * env_shape[1], (1. - elevation / np.pi) * env_shape[0]])
                pts = np.array(pts)
                df = np.diff(np.array(azs + [azs[0]]))
                cnd = np.bitwise_or(np.isclose(df, 0), df < 0)
                p_true, p_false = np.where(cnd)[0][0], np.where(~cnd)[0][0]    # IndexError -> the reference's except
                rows, cols = env_shape[:2]
                if np.sum(cnd) == 1:        # top
                    pts = np.vstack([pts[:p_true + 1], [cols, pts[p_true][1]], [cols, 0], [0, 0],
                                     [0, pts[np.mod(p_true + 1, N)][1]], pts[p_true + 1:]])
                elif np.sum(~cnd) == 1:     # bottom
                    pts = np.vstack([pts[:p_false + 1], [0, pts[p_false][1]], [0, rows], [cols, rows],
                                     [cols, pts[np.mod(p_false + 1, N)][1]], pts[p_false + 1:]])
                return np.array(pts), [], [], []
        except Exception:
            return [], [], [], []
