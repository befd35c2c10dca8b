"""k_tile_rows (round 6) renders rotate_bound -> flip -> resize(INTER_AREA) tiles (generator.py:163-170) by ROW WALKS: a
lane per canvas row, the horizontal folds of resizeArea_ in a register driven by a per-column table (rr_device.h
coltab_cell_pass1 / 2), cell sums through a small buffer, vertical folds per output pixel.  tests/hostemu runs the same
column table, walk rule, chunking and pair-texture fetch lane by lane on the CPU; here its tiles are compared with
raw_tile_pixel -- the one-thread-per-pixel definition the oracle tests pin -- bit for bit: on every rotate tile of a KITTI
scene, with a buffer so small that wide tiles are folded in column chunks and tall ones in many groups, and on plans whose
scales sit on the awkward values (cells that end within 1e-3 of a column edge, integer scale_x, the clamped last cell)."""
import ctypes

import numpy as np

import helpers as h

PLAN_INT = dict(status=0, kind=1, tex=2, flip=3, tw=4, th=5, nW=19, nH=20, rs_mode=21)


def _plans(sc, frame, H, W):
    lib = h.hostemu()
    drops = np.ascontiguousarray(sc.product_drops(frame))
    n = len(drops)
    psz = lib.emu_sizeof_plan()
    plans = np.zeros((n, psz), np.uint8)
    poly = np.zeros(n * 72, np.int32)
    npts = np.zeros(n, np.int32)
    sizes = np.zeros(n, np.int64)
    texels, hs, ws, offs = h.hb.pack_streak_db(sc.db.streaks_light)
    lib.emu_plan(h._p(drops), n, ctypes.byref(sc.cam), H, W, sc.He, sc.We, h._p(hs), h._p(ws), ctypes.c_double(1.0), h._p(plans), h._p(poly),
                 h._p(npts), h._p(sizes))
    return plans, sizes, (texels, hs, ws, offs)


def _run(lib, plan, db, buf):
    texels, hs, ws, offs = db
    ints = plan[:24 * 4].view(np.int32)
    tw, th = int(ints[PLAN_INT['tw']]), int(ints[PLAN_INT['th']])
    out = np.full(max(tw * th, 1), -1.0)
    ref = np.full(max(tw * th, 1), -2.0)
    rc = lib.emu_tile_rows(h._p(plan), h._p(texels), h._p(hs), h._p(ws), h._p(offs), int(buf), h._p(out), h._p(ref))
    return rc, out, ref


def test_row_walk_tiles_equal_the_definition_on_a_kitti_scene(tmp_path):
    H, W = 375, 1242
    sc = h.Scene(tmp_path, H, W, 1400, cam=h.KITTI, seed0=3000)
    plans, sizes, db = _plans(sc, 0, H, W)
    lib = h.hostemu()
    taken = chunked = 0
    for k in range(len(plans)):
        ints = plans[k][:24 * 4].view(np.int32)
        if ints[PLAN_INT['status']] != 0 or sizes[k] == 0 or ints[PLAN_INT['kind']] != 1:
            continue
        for buf in (344, 96):                          # the kernel's buffer; one that forces column chunks and one-row groups
            rc, out, ref = _run(lib, plans[k], db, buf)
            if rc == -1:
                continue
            assert rc == 0, (k, buf, rc, ints[[4, 5, 19, 20]])
            taken += 1
            chunked += buf == 96
    assert taken > 600 and chunked > 250, (taken, chunked)


def test_row_walk_on_awkward_scales(tmp_path):
    """Hand-made plans: the rotation of a real drop, the output size swept so that scale_x = nW / tw passes through
    integers, values a hair away from them, and small values near the kernel's limit of 2."""
    H, W = 375, 1242
    sc = h.Scene(tmp_path, H, W, 300, cam=h.KITTI, seed0=3100)
    plans, sizes, db = _plans(sc, 0, H, W)
    lib = h.hostemu()
    psz = plans.shape[1]
    # byte offsets of the fields the sweep rewrites (DropPlan, rr_device.h): tw, th at ints 4, 5; the four doubles scale_x,
    # scale_y, inv_sx, inv_sy are the LAST 32 bytes
    done = 0
    rot = [k for k in range(len(plans)) if plans[k][:96].view(np.int32)[PLAN_INT['status']] == 0 and sizes[k] > 0 and plans[k][:96].view(np.int32)[PLAN_INT['kind']] == 1]
    for k in rot[:40]:
        base = plans[k].copy()
        ints = base[:96].view(np.int32)
        nW, nH = int(ints[PLAN_INT['nW']]), int(ints[PLAN_INT['nH']])
        for tw in sorted(set([1, 2, 3, 5, 7, nW // 8, nW // 4, nW // 3, nW // 2, nW // 2 - 1, 31, 47, 64])):
            for th in (1, 2, 3, 9, nH // 7, nH // 2, nH - 1):
                if tw < 1 or th < 1 or tw > 64 or th > nH:
                    continue
                q = base.copy()
                qi = q[:96].view(np.int32)
                qi[PLAN_INT['tw']], qi[PLAN_INT['th']] = tw, th
                sc_ = q[psz - 32:].view(np.float64)
                inv_sx, inv_sy = tw / nW, th / nH
                sc_[:] = (1.0 / inv_sx, 1.0 / inv_sy, inv_sx, inv_sy)
                isx, isy = round(sc_[0]), round(sc_[1])
                eps = 2.220446049250313e-16
                if sc_[0] < 1 or sc_[1] < 1:
                    continue
                fast = abs(sc_[0] - isx) < eps and abs(sc_[1] - isy) < eps
                qi[PLAN_INT['rs_mode']] = 1 if fast else 0
                rc, out, ref = _run(lib, q, db, 344)
                if rc == -1:
                    continue
                assert rc == 0, (k, tw, th, nW, nH, rc)
                done += 1
    assert done > 1500, done
