"""Environment-map estimation pre-pass (reference common/bad_weather.py:707-853,
EnvironmentMapGenerator): cylindrical un-projection of the frame into a 360-degree lat-long
map.  It produces the `envmap` INPUT of the hot path and runs on the device (csrc/rr_prepass.h;
SURVEY 8f "next" #2).  This class builds the projection tables the kernels gather through -- they depend only on
(H, W, focal) and are cached; the reference's per-pixel Python fill loops (fill_matrices) are replaced by equivalent
vector operations -- and keeps the reference-signature call."""
import numpy as np

from . import imgops


class EnvironmentMapGenerator:
    def __init__(self, f, image_width, image_height):
        self.image_width = image_width
        self.image_height = image_height
        self.focal = int(((f * 1000) / 12.7) * image_width)          # bad_weather.py:712
        self._tables = {}

    def _max_min_x(self, center):
        s = self.focal
        max_x = round(s * np.arctan(center[0] / self.focal) + center[0])   # :730-740
        min_x = round(s * np.arctan(-center[0] / self.focal) + center[0])
        return int(max_x), int(min_x)

    def _projection(self, H, W):
        key = (H, W)
        if key in self._tables:
            return self._tables[key]
        center = np.array([int(W // 2), int(H // 2)])
        max_x, min_x = self._max_min_x(center)
        cw = int(max_x - min_x) + 1
        yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing='ij')
        d_row = yy - center[1]
        d_col = xx - center[0]
        rows = np.round(self.focal * (d_row / np.sqrt(d_col ** 2 + self.focal ** 2)) + center[1])      # :722-728
        cols = np.round(self.focal * np.arctan(d_col / self.focal) + center[0]) - min_x
        key_flat = rows.astype(np.int32).astype(np.int64).ravel() * cw + cols.astype(np.int32).astype(np.int64).ravel()
        uniq, first = np.unique(key_flat, return_index=True)                # first source pixel wins (:762)
        tab = dict(cw=cw, uniq=uniq, first=first)     # the column fills (fill_matrices :821-853) are derived on the device
        self._tables[key] = tab
        return tab

    def device_tables(self, H, W):
        """(cw, uniq, first) for rr_set_envmap_geometry."""
        t = self._projection(H, W)
        return t['cw'], t['uniq'].astype(np.int32), t['first'].astype(np.int32)

    def generate_map(self, background, hip=None):
        """The reference's call (bad_weather.py:742-819; generator.py:400) for one frame, on the device
        (rr_prepass_frames in RR_PRE_ENV_ONLY mode): background float BGR in [0,1] -> float BGR map of width
        We = cw + 2*(cw // 2).  Generator.run does not come through here (no host round trip there)."""
        from .. import hip_backend
        hip = hip or hip_backend.shared_context()
        H, W = background.shape[:2]
        hip.set_prepass_kernels(imgops.gaussian_kernel(25, 25), imgops.gaussian_kernel(15, 0))
        hip.set_envmap_geometry(H, W, *self.device_tables(H, W))
        return hip.env_maps([background])[0] / 255.0
