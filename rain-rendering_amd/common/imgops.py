"""Small image helpers for the host-side driver (I/O and the pre-pass), replacing the
handful of cv2 calls the reference makes OUTSIDE the hot path.  PARITY UNPINNED: OpenCV is
not installed here and the reference has no golden outputs; the algorithms follow OpenCV's
documented conventions (BORDER_REFLECT_101, half-pixel-centre bilinear resize)."""
import os

import numpy as np


def gaussian_kernel(ksize, sigma):
    """cv::getGaussianKernel."""
    if sigma <= 0:
        sigma = 0.3 * ((ksize - 1) * 0.5 - 1) + 0.8
    x = np.arange(ksize) - (ksize - 1) / 2.0
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return k / k.sum()


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
# deflate strategy of the scanline writer: 'fast' (the library's own run-length + dynamic-Huffman encoder: the size of
# zlib's Z_RLE at a third of its CPU time -- the driver is bound by the deflate of its two files per frame), 'rle' (zlib
# Z_RLE), 'default' (zlib LZ77: the files of the Python writer, byte for byte), 'huffman'.  Decoded pixels are the same
# whatever the choice.
PNG_STRATEGY = {'default': 0, 'rle': 1, 'huffman': 2, 'fast': 3}[os.environ.get('RAIN_PNG_STRATEGY', 'fast')]


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


def _native_png(path):
    """(lib, W, H, channels, depth) when the library's PNG reader takes the file, else None."""
    import ctypes
    from .. import hip_backend
    if not str(path).lower().endswith('.png'):
        return None
    lib = hip_backend.load_library()
    w, h, c, d = (ctypes.c_int32() for _ in range(4))
    if lib.rr_png_info(os.fsencode(path), ctypes.byref(w), ctypes.byref(h), ctypes.byref(c), ctypes.byref(d)) != 0:
        return None
    return lib, w.value, h.value, c.value, d.value


def imread_bgr(path, out=None):
    """cv2.imread(path): 8-bit, 3 channels, BGR order.  PNGs go through the library's reader (rr_png_read_bgr8: no
    interpreter lock, straight into `out` when given); anything it does not take through PIL."""
    info = _native_png(path)
    if info is not None and info[4] == 8:
        lib, w, h = info[:3]
        dst = out if out is not None else np.empty((h, w, 3), np.uint8)
        if dst.shape == (h, w, 3) and dst.dtype == np.uint8 and dst.flags['C_CONTIGUOUS']:
            if lib.rr_png_read_bgr8(os.fsencode(path), dst.ctypes.data, h, w) == 0:
                return dst
    from PIL import Image
    im = np.array(Image.open(path).convert('RGB'))
    return np.ascontiguousarray(im[..., ::-1])


def imread_unchanged(path):
    """cv2.imread(path, cv2.IMREAD_UNCHANGED) for 8/16-bit single-channel PNGs."""
    info = _native_png(path)
    if info is not None and info[3] == 1 and info[4] == 16:
        lib, w, h = info[:3]
        dst = np.empty((h, w), np.uint16)
        if lib.rr_png_read_gray16(os.fsencode(path), dst.ctypes.data, h, w) == 0:
            return dst
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


# matplotlib's default colour map (viridis) as plt.imsave uses it for a 2-D array: the 256 RGBA byte entries
# (cmap(arange(256)) * 255).astype(uint8), embedded so that the rain-mask PNGs do not depend on matplotlib being
# importable (tests/test_host_logic.py compares the table with matplotlib when it is).
_VIRIDIS_B64 = (
    "RAFU/0QCVf9EA1f/RQVY/0UGWv9FCFv/Rglc/0YLXv9GDF//Rg5h/0cPYv9HEWP/RxJl/0cUZv9HFWf/RxZp/0cYav9IGWv/SBps/0gcbv9IHW"
    "//SB5w/0ggcf9IIXL/SCJz/0gjdP9HJXX/RyZ2/0cnd/9HKHj/Ryp5/0crev9HLHv/Ri18/0YvfP9GMH3/RjF+/0Uyf/9FNH//RTWA/0U2gf9E"
    "N4H/RDmC/0M6g/9DO4P/QzyE/0I9hP9CPoX/QkCF/0FBhv9BQob/QEOH/0BEh/8/RYf/P0eI/z5IiP8+SYn/PUqJ/z1Lif89TIn/PE2K/zxOiv"
    "87UIr/O1GK/zpSi/86U4v/OVSL/zlVi/84Vov/OFeM/zdYjP83WYz/NlqM/zZbjP81XIz/NV2M/zRejf80X43/M2CN/zNhjf8yYo3/MmON/zFk"
    "jf8xZY3/MWaN/zBnjf8waI3/L2mN/y9qjf8ua47/LmyO/y5tjv8tbo7/LW+O/yxwjv8scY7/LHKO/ytzjv8rdI7/KnWO/yp2jv8qd47/KXiO/y"
    "l5jv8oeo7/KHqO/yh7jv8nfI7/J32O/yd+jv8mf47/JoCO/yaBjv8lgo7/JYON/ySEjf8khY3/JIaN/yOHjf8jiI3/I4mN/yKJjf8iio3/IouN"
    "/yGMjf8hjYz/IY6M/yCPjP8gkIz/IJGM/x+SjP8fk4v/H5SL/x+Vi/8flov/HpeK/x6Yiv8emYr/HpmK/x6aif8em4n/HpyJ/x6diP8enoj/Hp"
    "+I/x6gh/8foYf/H6KG/x+jhv8gpIX/IKWF/yGmhf8hp4T/IqeE/yOog/8jqYL/JKqC/yWrgf8mrIH/J62A/yiuf/8pr3//KrB+/yuxff8ssX3/"
    "LrJ8/y+ze/8wtHr/MrV6/zO2ef81t3j/Nrh3/zi5dv85uXb/O7p1/z27dP8+vHP/QL1y/0K+cf9EvnD/Rb9v/0fAbv9JwW3/S8Js/03Ca/9Pw2"
    "n/UcRo/1PFZ/9Vxmb/V8Zl/1nHZP9byGL/Xslh/2DJYP9iyl//ZMtd/2fMXP9pzFv/a81Z/23OWP9wzlb/cs9V/3TQVP930FL/edFR/3zST/9+"
    "0k7/gdNM/4PTS/+G1En/iNVH/4vVRv+N1kT/kNZD/5LXQf+V1z//l9g+/5rYPP+d2Tr/n9k4/6LaN/+l2jX/p9sz/6rbMv+t3DD/r9wu/7LdLP"
    "+13Sv/t90p/7reJ/+93ib/v98k/8LfIv/F3yH/x+Af/8rgHv/N4B3/z+Ec/9LhG//U4Rr/1+IZ/9riGP/c4hj/3+MY/+HjGP/k4xj/5+QZ/+nk"
    "Gf/s5Br/7uUb//HlHP/z5R7/9uYf//jmIf/65iL//eck/w=="
)
_viridis = None


def viridis_lut():
    """(256, 4) uint8 RGBA."""
    global _viridis
    if _viridis is None:
        import base64
        _viridis = np.frombuffer(base64.b64decode(_VIRIDIS_B64), np.uint8).reshape(256, 4).copy()
    return _viridis


def png_from_scanlines(path, rows, width, height, level=None, strategy=None):
    """An RGBA PNG file from its filtered scanlines (height rows of 1 + 4*width bytes, as the library's
    rr_frame_out.rainy_png / mask_png deliver them): zlib deflate + chunk framing inside the library
    (rr_png_write_scanlines), off the interpreter lock."""
    from .. import hip_backend
    rows = np.ascontiguousarray(rows, np.uint8)
    assert rows.nbytes == height * (1 + 4 * width)
    rc = hip_backend.load_library().rr_png_write_scanlines(os.fsencode(path), rows.ctypes.data, int(width), int(height),
                                                           PNG_LEVEL if level is None else int(level),
                                                           PNG_STRATEGY if strategy is None else int(strategy))
    if rc != 0:
        raise IOError("rr_png_write_scanlines(%s) failed (%d)" % (path, rc))


def imsave_scalar(path, a):
    """plt.imsave(path, 2-D float array): min/max normalised, viridis colour map, RGBA
    (reference generator.py:467)."""
    a = np.asarray(a, np.float64)
    lo, hi = float(a.min()), float(a.max())
    norm = np.zeros_like(a) if hi <= lo else (a - lo) / (hi - lo)
    idx = np.clip((norm * 256).astype(np.int64), 0, 255)
    _save_rgba(path, np.ascontiguousarray(viridis_lut()[idx]))
