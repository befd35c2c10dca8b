"""CPU tier: the device's PNG entropy coder (csrc/rr_deflate.h, the code k_pngz_blocks / k_pngz_pack run) built for the host
by tests/hostemu: zlib inflates what it writes -- every byte back, Adler-32 checked by zlib itself."""
import ctypes
import zlib

import numpy as np
import pytest

import helpers as h

MAGIC = b'RRZ1'


def pngz(rows):
    emu = h.hostemu()
    rows = np.ascontiguousarray(rows, np.uint8).ravel()
    dst = np.zeros_like(rows)
    emu.emu_pngz.restype = ctypes.c_int64
    emu.emu_pngz.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    total = emu.emu_pngz(rows.ctypes.data_as(ctypes.c_void_p), rows.size, dst.ctypes.data_as(ctypes.c_void_p))
    return int(total), dst


def scanlines(img_rgba):
    """Sub-filtered scanlines of an RGBA image (what k_png_image writes): filter byte 1 + byte-wise differences."""
    H, W, _ = img_rgba.shape
    d = img_rgba.astype(np.int16)
    d[:, 1:] -= img_rgba[:, :-1].astype(np.int16)
    rows = np.empty((H, 1 + 4 * W), np.uint8)
    rows[:, 0] = 1
    rows[:, 1:] = (d & 255).astype(np.uint8).reshape(H, 4 * W)
    return rows


def check(rows):
    rows = np.ascontiguousarray(rows, np.uint8).ravel()
    total, dst = pngz(rows)
    if total == 0:
        assert np.array_equal(dst, rows)                     # did not fit: the scanlines stay
        return 0
    assert dst[:4].tobytes() == MAGIC and int(dst[4:8].view(np.uint32)[0]) == total and 16 + total <= rows.size
    back = zlib.decompress(dst[16:16 + total].tobytes())     # (checks the Adler-32 as well)
    assert back == rows.tobytes()
    return total


@pytest.mark.parametrize("H,W", [(37, 64), (96, 160), (375, 1242)])
def test_image_like_rows_round_trip(H, W):
    img = np.zeros((H, W, 4), np.uint8)
    img[..., :3] = (h.synthetic.make_frame(3, H, W) * 255).astype(np.uint8)
    img[..., 3] = 255
    rows = scanlines(img)
    total = check(rows)
    assert 0 < total < 0.8 * rows.size                       # residuals of a smooth image: well below the raw size


def test_constant_mask_rows_shrink_to_a_few_percent():
    H, W = 375, 1242
    img = np.zeros((H, W, 4), np.uint8)
    img[...] = (68, 1, 84, 255)                               # viridis(0)
    img[100:110, 200:260] = (253, 231, 37, 255)
    total = check(scanlines(img))
    assert 0 < total < 0.03 * H * (1 + 4 * W)


@pytest.mark.parametrize("n", [1, 2, 3, 127, 128, 129, 32767, 32768, 32769, 65536 + 5, 3 * 32768])
def test_sizes_around_the_span_and_block_edges(n):
    rng = np.random.RandomState(n)
    rows = (rng.randint(0, 4, n) * rng.randint(0, 2, n)).astype(np.uint8)      # small alphabet with zero runs
    check(rows)
    check(np.zeros(n, np.uint8))
    check(np.full(n, 255, np.uint8))


def test_incompressible_rows_are_left_alone_or_stored():
    rng = np.random.RandomState(1)
    rows = rng.randint(0, 256, 200000).astype(np.uint8)
    assert check(rows) == 0                                    # stored blocks + header do not fit: scanlines kept
    mixed = rows.copy()
    mixed[50000:] = 0                                          # one incompressible (stored) block, the rest shrinks
    assert check(mixed) > 0


def test_skewed_histograms_hit_the_length_limit():
    """Counts like Fibonacci numbers make the unrestricted Huffman tree deeper than 15: the count-based fix-up must leave a
    complete prefix code that zlib accepts."""
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    parts = [np.full(c, v, np.uint8) for v, c in enumerate(fib)]
    rng = np.random.RandomState(2)
    rows = np.concatenate(parts)
    rng.shuffle(rows)                                          # (no runs: shuffled)
    rows = rows[:32768]
    check(rows)
    check(np.concatenate([rows, rows[::-1], rows]))
