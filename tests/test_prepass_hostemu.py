"""CPU tier: the per-pixel pre-pass arithmetic of rr_prepass.h (compiled for the host by
tests/hostemu) against the numpy oracle oracle/prepass.py, and the host mirrors
(common/add_attenuation.py, common/envmap.py) against the same oracle."""
import ctypes
import importlib

import numpy as np
import pytest

import helpers as h
from oracle import prepass as op

fogmod = importlib.import_module('rain-rendering_amd.common.add_attenuation')
envmod = importlib.import_module('rain-rendering_amd.common.envmap')


def scene(H, W, seed, dtype=np.float32):
    bg = h.synthetic.make_frame(seed, H, W)
    rng = np.random.RandomState(seed)
    depth = (np.linspace(80, 2, H)[:, None] * np.ones((1, W)) + rng.uniform(0, 3, (H, W))).astype(dtype)
    return bg, depth


def emu_prepass(bg, depth, rain, focal_m=0.006, f_number=6.0, exposure=2, gain=20):
    emu = h.hostemu()
    H, W = bg.shape[:2]
    fog = fogmod.FogRain(rain_intensity=rain, focal=focal_m, f_number=f_number, angle=90, exposure=exposure, camera_gain=gain)
    be, bh, num, den = fog.constants()
    gen = envmod.EnvironmentMapGenerator(focal_m, W, H)
    cw, uniq, first = gen.device_tables(H, W)
    We = cw + 2 * (cw // 2)
    fw = np.ascontiguousarray(op.gaussian_kernel(25, 25))
    ew = np.ascontiguousarray(op.gaussian_kernel(15, 0))
    rainy = np.zeros((H, W, 3))
    env = np.zeros((H, We, 3))
    env8 = np.zeros((H, We, 3), np.uint8)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    emu.emu_prepass.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int] + \
        [ctypes.c_double] * 4 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + \
        [ctypes.c_void_p] * 5
    rc = emu.emu_prepass(H, W, p(bg), p(depth), int(depth.dtype == np.float64), be, bh, num, den, 25, p(fw), 15, p(ew),
                         cw, len(uniq), p(uniq), p(first), p(rainy), p(env), p(env8))
    assert rc == We
    return rainy, env, env8


@pytest.mark.parametrize("H,W,dtype", [(96, 160, np.float32), (75, 131, np.float64), (40, 64, np.float32)])
def test_prepass_arithmetic_matches_oracle(H, W, dtype):
    bg, depth = scene(H, W, 5, dtype)
    rainy, env, env8 = emu_prepass(bg, depth, 50)
    want = op.fog_rain_layer(bg, depth, 50, 6.0, 2, 20)
    # expf/exp come from different libms and the channel mean is summed in another order
    assert np.abs(rainy - want).max() < 2e-7
    # environment map from the emulator's OWN fog output, so that only the map logic is compared
    e_bgr = op.generate_env_map(rainy, 0.006)
    want8 = np.rint(e_bgr * 255).astype(np.uint8)
    assert env8.shape == want8.shape
    assert np.array_equal(env8, want8)
    want_xyY = op.env_to_xyY(e_bgr)
    assert np.abs(env - want_xyY).max() < 1e-12


def test_host_constants_equal_oracle():
    """The host side of the pre-pass: scalar constants and projection tables (the arithmetic itself lives on the device;
    the reference-signature calls FogRain.fog_rain_layer / generate_map are covered on the GPU tier)."""
    be, bh, num, den = fogmod.FogRain(25, 0.006, 6.0, 90, 2, 20).constants()
    obe, obh, oscale = op.fog_constants(25, 6.0, 2, 20)
    assert be == obe and bh == obh and abs(num / den - oscale) <= 1e-12 * oscale
    cw, uniq, first = envmod.EnvironmentMapGenerator(0.006, 160, 96).device_tables(96, 160)
    ocw, ouniq, ofirst = op.env_geometry(0.006, 96, 160)
    assert cw == ocw and np.array_equal(uniq, ouniq) and np.array_equal(first, ofirst)
