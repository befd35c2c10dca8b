"""CPU tier: the defocus-blur work split (rr_device.h: blur_is_small / blur_layout, compiled for the host by
tests/hostemu) swept over tile shapes and radii: every sub-tile the blur kernels would stage must fit the LDS capacities
they are launched with, for each of the three capacity presets (RR_OPT_BLUR_WORKGROUPS 3 / 4 / 5)."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PRESETS = ((3072, 2048), (2816, 2048), (2304, 1600))       # rainhip.hip: enqueue(), blur_wg = 3, 4, 5


def _lib():
    import __graft_entry__ as ge
    ge.build()
    lib = ctypes.CDLL(os.path.join(HERE, 'hostemu', 'libhostemu.so'))
    lib.emu_blur_layout.argtypes = [ctypes.c_int] * 8 + [ctypes.c_void_p]
    lib.emu_blur_layout.restype = ctypes.c_int
    return lib


def test_every_subtile_fits_the_lds_capacities():
    lib = _lib()
    out = np.zeros(4, np.int32)
    rng = np.random.RandomState(11)
    cases = []
    for tw in (1, 2, 3, 5, 8, 13, 17, 32, 52, 85, 200, 640):
        for th in (1, 4, 11, 30, 64, 150, 400, 1000):
            for r1 in (1, 2, 5, 9, 16, 31, 32, 47, 48, 49, 200):
                for r2 in sorted({0, 1, r1 // 2, max(r1 // 2 - 1, 0), min(r1, 48)}):
                    cases.append((tw, th, r1, r2))
    for _ in range(20000):
        r1 = int(rng.randint(1, 60))
        cases.append((int(rng.randint(1, 120)), int(rng.randint(1, 500)), r1, int(rng.randint(0, r1 + 1))))
    for _ in range(5000):                                    # the small shapes the wave-per-drop kernel takes
        r1 = int(rng.randint(1, 14))
        cases.append((int(rng.randint(1, 14)), int(rng.randint(1, 44)), r1, int(rng.randint(0, r1 + 1))))
    n_small = n_fused = n_slow = 0
    for (tw, th, r1, r2) in cases:
        ew, eh = tw + 2 * r2, th + 2 * r1
        for bx, by in PRESETS:
            bad = lib.emu_blur_layout(ew, eh, r1, r2, tw, th, bx, by, out.ctypes.data)
            assert bad == 0, (tw, th, r1, r2, bx, by, out.tolist(), bad)
            small, fused, wo, ho = out.tolist()
            if r1 > 48:
                assert not fused and not small, (tw, th, r1, r2)
            if fused and not small:
                assert 1 <= wo <= ew and 1 <= ho <= eh
        n_small += small
        n_fused += bool(fused and not small)
        n_slow += bool(not fused and not small)
    assert n_small > 1000 and n_fused > 1000 and n_slow > 100          # the sweep reaches all three kernels


def test_kitti_shaped_drops_mostly_take_one_band():
    """The shapes the 100 mm/hr KITTI workload is made of (raw tile ~17x30, radii ~10 / 5): the default capacities take
    the median drop in one piece."""
    lib = _lib()
    out = np.zeros(4, np.int32)
    assert lib.emu_blur_layout(15 + 10, 27 + 18, 9, 5, 15, 27, 2816, 2048, out.ctypes.data) == 0
    small, fused, wo, ho = out.tolist()
    assert (small, fused) == (0, 1) and (wo, ho) == (25, 45)
