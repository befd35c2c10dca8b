"""Environment-map estimation pre-pass (reference common/bad_weather.py:707-853,
EnvironmentMapGenerator): cylindrical un-projection of the frame into a 360-degree lat-long
map.  It produces the `envmap` INPUT of the hot path and runs on the host (SURVEY 8f "next"
#2).  The projection tables depend only on (H, W, focal) and are cached; the reference's
per-pixel Python fill loops (fill_matrices) are replaced by equivalent vector operations."""
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
        mask = np.zeros((H, cw), np.uint8)
        mask.reshape(-1)[uniq] = 255
        tab = dict(cw=cw, uniq=uniq, first=first, mask=mask)
        # column-fill tables (fill_matrices :821-853): first filled row from the top / from the bottom
        half = H // 2
        top = mask[:half]
        tab['top_unfilled'] = np.nonzero(top == 0)
        tab['top_src_row'] = np.argmax(top > 0, axis=0)
        bot = mask[::-1][:half]
        tab['bot_unfilled'] = np.nonzero(bot == 0)
        tab['bot_src_row'] = np.argmax(mask[half:][::-1] > 0, axis=0)
        self._tables[key] = tab
        return tab

    def device_tables(self, H, W):
        """(cw, uniq, first) for rr_set_envmap_geometry."""
        t = self._projection(H, W)
        return t['cw'], t['uniq'].astype(np.int32), t['first'].astype(np.int32)

    def generate_map(self, background):
        """reference bad_weather.py:742-819; background is float BGR in [0,1]."""
        bg8 = (background * 255).astype(np.uint8)
        H, W = bg8.shape[:2]
        t = self._projection(H, W)
        cw = t['cw']
        cyl = np.zeros((H, cw, 3), np.uint8)
        cyl.reshape(-1, 3)[t['uniq']] = bg8.reshape(-1, 3)[t['first']]
        mask = t['mask']
        half = H // 2
        # bottom half: every unfilled pixel takes its column's first filled pixel seen from the bottom
        fl = cyl[::-1]
        tmp = fl[:half].copy()
        r, c = t['bot_unfilled']
        tmp[r, c] = fl[t['bot_src_row'][c], c]
        cyl[-half:] = tmp[::-1] if half else cyl[-half:]
        # top half
        r, c = t['top_unfilled']
        cyl[r, c] = cyl[t['top_src_row'][c], c]
        lw = int(cw / 2)
        result = np.zeros((H, cw + 2 * lw, 3), np.uint8)
        result[:, lw:lw + cw] = cyl
        mres = np.zeros((H, cw + 2 * lw), np.uint8)
        mres[:, lw:lw + cw] = mask
        side = cyl[:, 0:lw][:, ::-1]
        result[:, 0:side.shape[1]] = side
        mside = mask[:, :cw // 2][:, ::-1]
        mres[:, :mside.shape[1]] = mside
        side = cyl[:, cw // 2:][:, ::-1]
        result[:, result.shape[1] - side.shape[1]:] = side
        mside = mask[:, cw // 2:][:, ::-1]
        mres[:, mres.shape[1] - side.shape[1]:] = mside
        blur = imgops.gaussian_blur_u8(result, 15, 0)                       # :815
        result = np.where(mres[..., None] == 0, blur, result)               # :816-817
        return result / 255.0
