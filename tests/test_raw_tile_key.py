"""CPU tier: the 32-byte raw-tile key k_dedup compares (rr_device.h raw_tile_key, round 6) against the plan fields the tile
kernels read.  Equal keys must mean equal tile parameters -- otherwise two drops would share a tile that is not the same
bits -- and the key should lose (next to) none of the sharing the comparison of the derived fields found."""
import ctypes
import numpy as np
import pytest

import helpers as h
import test_homography_rounding as thr


@pytest.mark.parametrize("noise", [0.0, 3.0])
def test_equal_keys_mean_equal_tile_parameters(tmp_path, noise):
    sc = h.Scene(tmp_path, 375, 1242, 2048, n_frames=3, cam=h.KITTI, seed0=5100)
    emu = h.hostemu()
    texels, hs, ws, offs = h.hb.pack_streak_db(sc.db.streaks_light)
    PLAN = thr.PLAN_DTYPE
    drops = np.concatenate([np.ascontiguousarray(sc.product_drops(i, noise_std=noise, noise_scale=1.0)) for i in range(3)])
    n = len(drops)
    plans = np.zeros(n, PLAN); poly = np.zeros(n * 72, np.int32); npts = np.zeros(n, np.int32); sizes = np.zeros(n, np.int64)
    emu.emu_plan(h._p(drops), n, ctypes.byref(sc.cam), 375, 1242, sc.He, sc.We, h._p(hs), h._p(ws), ctypes.c_double(1.0),
                 h._p(plans), h._p(poly), h._p(npts), h._p(sizes))
    keys = np.zeros((n, 8), np.uint32)
    emu.emu_raw_tile_keys(h._p(drops), n, h._p(plans), h._p(keys))
    ok = plans['status'] == 0
    assert (keys[ok, 0] != 0xffffffff).all() and (keys[~ok, 0] == 0xffffffff).all()
    # the plan fields k_tile_* read (what round 5's k_dedup compared), as bytes
    fields = ['kind', 'tex', 'flip', 'tw', 'th', 'bw0', 'nW', 'nH', 'rs_mode', 'isx', 'isy', 'mi', 'ma', 'scale_x', 'scale_y', 'inv_sx', 'inv_sy']
    def tile_bytes(i):
        return b''.join(np.ascontiguousarray(plans[f][i]).tobytes() for f in fields)
    by_key, by_fields = {}, {}
    for i in np.nonzero(ok)[0]:
        by_key.setdefault(keys[i].tobytes(), []).append(i)
        by_fields.setdefault(tile_bytes(i), []).append(i)
    for members in by_key.values():
        first = tile_bytes(members[0])
        assert all(tile_bytes(m) == first for m in members[1:])
    # sharing found by the key vs by the derived fields
    n_ok = int(ok.sum())
    share_key, share_fields = 1 - len(by_key) / n_ok, 1 - len(by_fields) / n_ok
    assert share_key >= share_fields - 0.002, (share_key, share_fields)
    if noise == 0.0:
        assert share_key > 0.3
