"""CPU tier: the input files as filtered scanlines (rr_io_read_frames_rows) and the rule the device reverses their filters
with (csrc/rr_pngrows.h, built for the host by tests/hostemu) against the library's own readers and PIL."""
import ctypes
import os

import numpy as np
import pytest

import helpers as h

hb = h.hb


def _write(tmp, H, W, seed, mode):
    """An 8-bit RGB image and a 16-bit gray depth map written by PIL (adaptive filters: every filter type occurs)."""
    from PIL import Image
    rng = np.random.RandomState(seed)
    img = (h.synthetic.make_frame(seed, H, W)[..., ::-1] * 255).astype(np.uint8)
    if mode == 'noise':
        img = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
    d16 = (np.linspace(80, 2, H)[:, None] * np.ones((1, W)) * 256 + rng.uniform(0, 700, (H, W))).astype(np.uint16)
    ip, dp = os.path.join(tmp, 'i%d.png' % seed), os.path.join(tmp, 'd%d.png' % seed)
    Image.fromarray(img).save(ip, compress_level=1 if mode != 'noise' else 6)
    Image.fromarray(d16).save(dp)
    return ip, dp, img[..., ::-1].copy(), d16


def _rows(paths_i, paths_d, H, W):
    n = len(paths_i)
    ri = np.zeros((n, H * (1 + 3 * W)), np.uint8)
    rd = np.zeros((n, H * (1 + 2 * W)), np.uint8)
    st = hb.io_read_frames_rows(paths_i, paths_d, H, W, ri, rd)
    return st, ri, rd


def _unfilter(rows, H, W, bpp):
    emu = h.hostemu()
    out = np.zeros((H, W, 3), np.uint8) if bpp == 3 else np.zeros((H, W), np.uint16)
    emu.emu_png_unfilter.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    assert emu.emu_png_unfilter(rows.ctypes.data, H, W, bpp, out.ctypes.data) == 0
    return out


@pytest.mark.parametrize("H,W,mode", [(48, 80, 'smooth'), (37, 131, 'noise'), (375, 1242, 'smooth')])
def test_rows_reader_and_unfilter_rule_give_the_decoders_pixels(tmp_path, H, W, mode):
    files = [_write(str(tmp_path), H, W, s, mode) for s in (1, 2)]
    st, ri, rd = _rows([f[0] for f in files], [f[1] for f in files], H, W)
    assert (st == 0).all()
    kinds = set()
    for k, (ip, dp, bgr, d16) in enumerate(files):
        kinds |= set(ri[k].reshape(H, 1 + 3 * W)[:, 0].tolist()) | set(rd[k].reshape(H, 1 + 2 * W)[:, 0].tolist())
        assert np.array_equal(_unfilter(ri[k], H, W, 3), bgr)                       # what PIL wrote
        assert np.array_equal(_unfilter(rd[k], H, W, 2), d16)
        # and what the library's full readers deliver
        bg_block, d_block = np.zeros((1, H * W * 3), np.uint8), np.zeros((1, H * W * 2), np.uint8)
        assert (hb.io_read_frames([ip], [dp], H, W, bg_block, d_block, depth_u16=True) == 0).all()
        assert np.array_equal(bg_block.reshape(H, W, 3), bgr) and np.array_equal(d_block.view(np.uint16).reshape(H, W), d16)
    assert len(kinds) >= 3, kinds                             # (PIL's adaptive filtering: several filter types were exercised)


def test_other_kinds_of_files_come_as_rows_of_filter_type_zero(tmp_path):
    """An RGBA image, a gray image, an 8-bit depth map: not what the row format carries -- the reader decodes them on the host
    and hands over rows of filter type 0 in PNG sample order (or reports what the full readers report)."""
    from PIL import Image
    H, W = 40, 64
    rng = np.random.RandomState(3)
    rgba = rng.randint(0, 256, (H, W, 4)).astype(np.uint8)
    gray = rng.randint(0, 256, (H, W)).astype(np.uint8)
    d16 = rng.randint(0, 65536, (H, W)).astype(np.uint16)
    pa, pg, pd = [str(tmp_path / n) for n in ('a.png', 'g.png', 'd.png')]
    Image.fromarray(rgba).save(pa)
    Image.fromarray(gray).save(pg)
    Image.fromarray(d16).save(pd)
    st, ri, rd = _rows([pa, pg], [pd, pd], H, W)
    assert (st == 0).all()
    assert (ri.reshape(2, H, 1 + 3 * W)[:, :, 0] == 0).all()
    assert np.array_equal(_unfilter(ri[0], H, W, 3), rgba[..., 2::-1])
    assert np.array_equal(_unfilter(ri[1], H, W, 3), np.repeat(gray[..., None], 3, -1))
    assert np.array_equal(ri[0], hb.png_rows_of(np.ascontiguousarray(rgba[..., 2::-1])))
    assert np.array_equal(rd[0], rd[1]) and np.array_equal(_unfilter(rd[0], H, W, 2), d16)
    st, _, _ = _rows([pa], [pg], H, W)                        # an 8-bit file where the 16-bit depth map is expected
    assert st[0] != 0
    st, _, _ = _rows([str(tmp_path / 'missing.png')], [pd], H, W)
    assert st[0] != 0
