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
        t = StreakTable(0)
        for k in self.FIELDS:
            setattr(t, k, getattr(self, k)[idx])
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
        its call site passes use_pickle=False, generator.py:281; not implemented)."""
        print('Reading particles file {}'.format(self.streaks_path_xml))
        if not os.path.exists(self.streaks_path_xml):
            print("No existing path for XML file (" + str(self.streaks_path_xml) + ")")
            sys.exit(-1)
        try:
            simulation = parse(self.streaks_path_xml).getroot()
        except Exception:
            raise Exception("Reading XML file {} crashed, which is likely due to corrupted particles simulation "
                            "files. If so, delete this simulation folder manually and re-run to allow generation "
                            "of new simulation.".format(self.streaks_path_xml))
        rs = settings["render_scale"]
        gan = dataset == 'nuscenes_gan'
        r_gan = np.mean((image_shape_WH[0] / 1600, image_shape_WH[1] / 900)) if gan else None
        try:
            for frame in simulation:
                f = Frame()
                f.id = int(frame.attrib['id'])
                f.exposure_time = int(frame.attrib['t'])
                f.starting_time = int(frame.attrib['d'])
                f.streaks_count = int(frame.attrib['rs'])
                drops = list(frame)
                n = len(drops)
                # all drops of a frame are converted in bulk (one strtod pass per attribute instead of
                # ~20 Python float() calls per drop); same correctly rounded doubles
                att = [d.attrib for d in drops]
                pid = np.array([int(a["pid"]) for a in att], np.int64).reshape(n)
                wps = _bulk_vec([a["wp1"] for a in att], 3)
                wpe = _bulk_vec([a["wp2"] for a in att], 3)
                wd = np.stack([_bulk_vec([a['wd1'] for a in att], 1)[:, 0], _bulk_vec([a['wd2'] for a in att], 1)[:, 0]], axis=1) \
                    if n else np.zeros((0, 2))
                ip1 = _bulk_vec([a["ip1"] for a in att], 2)
                ip2 = _bulk_vec([a["ip2"] for a in att], 2)
                iw = np.stack([_bulk_vec([a['iw1'] for a in att], 1)[:, 0], _bulk_vec([a['iw2'] for a in att], 1)[:, 0]], axis=1) \
                    if n else np.zeros((0, 2))
                if gan:
                    ips, ipe, iws = ip1 * r_gan, ip2 * r_gan, iw * r_gan
                else:
                    ips, ipe, iws = ip1 / rs, ip2 / rs, iw / rs
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
                # dict.update semantics: a repeated pid keeps its FIRST position, LAST value
                order = {}
                for k in np.nonzero(keep)[0]:
                    order[int(pid[k])] = k
                idx = np.fromiter(order.values(), dtype=np.int64, count=len(order))
                t = StreakTable(len(idx))
                t.pid[:] = pid[idx]
                t.wps[:] = wps[idx]
                t.wpe[:] = wpe[idx]
                t.wd1[:] = wd[idx, 0]
                t.wd2[:] = wd[idx, 1]
                t.ips[:] = ips_i[idx]
                t.ipe[:] = ipe_i[idx]
                t.iw1[:] = iws[idx, 0]
                t.iw2[:] = iws[idx, 1]
                t.ratio[:] = ratio[idx]
                t.max_width[:] = max_width[idx]
                t.length[:] = length[idx]
                t.type[:] = dtype[idx]
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
