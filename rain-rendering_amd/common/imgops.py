"""Small image helpers for the host-side driver (I/O and the pre-pass), replacing the
handful of cv2 calls the reference makes OUTSIDE the hot path.  PARITY UNPINNED: OpenCV is
not installed here and the reference has no golden outputs; the algorithms follow OpenCV's
documented conventions (BORDER_REFLECT_101, half-pixel-centre bilinear resize)."""
import os

import numpy as np
from scipy.ndimage import correlate1d


def gaussian_kernel(ksize, sigma):
    """cv::getGaussianKernel."""
    if sigma <= 0:
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = np.arange(ksize) - (ksize - 1) / 2.0
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return k / k.sum()


def gaussian_blur(img, ksize, sigma):
    """cv2.GaussianBlur(img, (ksize, ksize), sigma) for float images, BORDER_REFLECT_101."""
    k = gaussian_kernel(ksize, sigma)
    out = correlate1d(img, k, axis=1, mode='mirror')
    return correlate1d(out, k, axis=0, mode='mirror')


def gaussian_blur_u8(img, ksize, sigma=0):
    out = gaussian_blur(img.astype(np.float64), ksize, sigma)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def resize_linear(img, dw, dh):
    """cv2.resize(img, (dw, dh)) (INTER_LINEAR, half-pixel centres, edge clamp) for float images."""
    sh, sw = img.shape[:2]
    if (sh, sw) == (dh, dw):
        return img.copy()

    def coords(d, s):
        f = (np.arange(d) + 0.5) * (s / d) - 0.5
        i0 = np.floor(f).astype(np.int64)
        w = f - i0
        i1 = i0 + 1
        w = np.where(i0 < 0, 0.0, w)
        i0c = np.clip(i0, 0, s - 1)
        i1c = np.clip(i1, 0, s - 1)
        return i0c, i1c, w

    y0, y1, wy = coords(dh, sh)
    x0, x1, wx = coords(dw, sw)
    a = img.astype(np.float64)
    shape_x = (1, dw) + (1,) * (a.ndim - 2)
    shape_y = (dh, 1) + (1,) * (a.ndim - 2)
    top = a[y0][:, x0] * (1 - wx.reshape(shape_x)) + a[y0][:, x1] * wx.reshape(shape_x)
    bot = a[y1][:, x0] * (1 - wx.reshape(shape_x)) + a[y1][:, x1] * wx.reshape(shape_x)
    return top * (1 - wy.reshape(shape_y)) + bot * wy.reshape(shape_y)


# zlib level of the PNGs the driver writes.  Pixels are what parity is about; matplotlib's default (6)
# costs ~0.4 s per 1242x375 RGBA frame, level 1 a quarter of that for ~15% larger files.
PNG_LEVEL = int(os.environ.get('RAIN_PNG_LEVEL', '1'))


def write_png_rgba(path, rgba, level=None):
    """8-bit RGBA PNG with the Sub filter on every row and one zlib stream: pixel-identical to what PIL (and
    therefore plt.imsave) decodes, in about half the time of PIL's adaptive filtering.  zlib releases the GIL,
    so the driver's I/O threads scale."""
    import struct
    import zlib
    rgba = np.ascontiguousarray(rgba, np.uint8)
    hh, ww = rgba.shape[:2]
    assert rgba.shape == (hh, ww, 4)
    flat = rgba.reshape(hh, -1)
    rows = np.empty((hh, 1 + ww * 4), np.uint8)
    rows[:, 0] = 1                                  # filter type 1 (Sub): byte - byte of the pixel to the left
    rows[:, 1:5] = flat[:, :4]
    rows[:, 5:] = flat[:, 4:] - flat[:, :-4]        # uint8 arithmetic wraps modulo 256, as the filter requires

    def chunk(tag, data):
        return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)

    blob = (b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', ww, hh, 8, 6, 0, 0, 0)) +
            chunk(b'IDAT', zlib.compress(rows.tobytes(), PNG_LEVEL if level is None else level)) + chunk(b'IEND', b''))
    with open(path, 'wb') as fh:
        fh.write(blob)


def _save_rgba(path, rgba):
    if os.environ.get('RAIN_PNG_WRITER', 'fast') == 'pil':
        from PIL import Image
        Image.fromarray(rgba, 'RGBA').save(path, compress_level=PNG_LEVEL)
    else:
        write_png_rgba(path, rgba)


def imread_bgr(path):
    """cv2.imread(path): 8-bit, 3 channels, BGR order."""
    from PIL import Image
    im = np.array(Image.open(path).convert('RGB'))
    return np.ascontiguousarray(im[..., ::-1])


def imread_unchanged(path):
    """cv2.imread(path, cv2.IMREAD_UNCHANGED) for 8/16-bit single-channel PNGs."""
    from PIL import Image
    try:
        return np.array(Image.open(path))
    except Exception:
        return None


def imsave_rgb(path, rgb_u8):
    """plt.imsave(path, float_rgb) equivalent for an already quantised image: RGBA PNG, alpha 255."""
    from PIL import Image
    h, w = rgb_u8.shape[:2]
    rgba = np.empty((h, w, 4), np.uint8)
    rgba[..., :3] = rgb_u8
    rgba[..., 3] = 255
    _save_rgba(path, rgba)


_viridis = None


def imsave_scalar(path, a):
    """plt.imsave(path, 2-D float array): min/max normalised, viridis colour map, RGBA
    (reference generator.py:467)."""
    global _viridis
    from PIL import Image
    a = np.asarray(a, np.float64)
    lo, hi = float(a.min()), float(a.max())
    norm = np.zeros_like(a) if hi <= lo else (a - lo) / (hi - lo)
    if _viridis is None:
        try:
            import matplotlib
            cmap = matplotlib.colormaps['viridis'] if hasattr(matplotlib, 'colormaps') else matplotlib.cm.get_cmap('viridis', 256)
            _viridis = (np.asarray(cmap(np.arange(256))) * 255).astype(np.uint8)
        except Exception:
            g = np.arange(256, dtype=np.uint8)
            _viridis = np.stack([g, g, g, np.full(256, 255, np.uint8)], axis=1)
    idx = np.clip((norm * 256).astype(np.int64), 0, 255)
    _save_rgba(path, np.ascontiguousarray(_viridis[idx]))
