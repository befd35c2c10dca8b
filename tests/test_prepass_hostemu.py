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


PRE_BG_F32, PRE_BG_U8, PRE_RAINY_F32, PRE_ENV_F32, PRE_DEPTH_U16 = 1, 2, 4, 8, 16         # rrpre::PRE_* (csrc/rr_prepass.h)


def emu_prepass(bg, depth, rain, focal_m=0.006, f_number=6.0, exposure=2, gain=20, tiled=1, seg_rows=64, narrow=False, fog_taps=25):
    """bg: float64 / float32 / uint8 image; narrow: float32 rainy_bg and xyY map (the float64 results rounded once)."""
    emu = h.hostemu()
    H, W = bg.shape[:2]
    fog = fogmod.FogRain(rain_intensity=rain, focal=focal_m, f_number=f_number, angle=90, exposure=exposure, camera_gain=gain)
    be, bh, num, den = fog.constants()
    gen = envmod.EnvironmentMapGenerator(focal_m, W, H)
    cw, uniq, first = gen.device_tables(H, W)
    We = cw + 2 * (cw // 2)
    fw = np.ascontiguousarray(op.gaussian_kernel(fog_taps, fog_taps))
    ew = np.ascontiguousarray(op.gaussian_kernel(15, 0))
    out_t = np.float32 if narrow else np.float64
    rainy = np.zeros((H, W, 3), out_t)
    env = np.zeros((H, We, 3), out_t)
    env8 = np.zeros((H, We, 3), np.uint8)
    bg = np.ascontiguousarray(bg)
    types = {np.dtype(np.float64): 0, np.dtype(np.float32): PRE_BG_F32, np.dtype(np.uint8): PRE_BG_U8}[bg.dtype]
    types |= (PRE_RAINY_F32 | PRE_ENV_F32) if narrow else 0
    types |= PRE_DEPTH_U16 if depth.dtype == np.uint16 else 0
    depth = np.ascontiguousarray(depth)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    emu.emu_prepass.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int] + \
        [ctypes.c_double] * 4 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + \
        [ctypes.c_void_p] * 5 + [ctypes.c_int] * 3
    rc = emu.emu_prepass(H, W, p(bg), p(depth), int(depth.dtype == np.float64), be, bh, num, den, fog_taps, p(fw), 15, p(ew),
                         cw, len(uniq), p(uniq), p(first), p(rainy), p(env), p(env8), types, int(tiled), int(seg_rows))
    assert rc == We, rc
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


@pytest.mark.parametrize("H,W,dtype,seg", [(96, 160, np.float32, 64), (75, 131, np.float64, 24), (40, 64, np.float32, 8), (9, 5, np.float32, 16),
                                           (200, 67, np.float32, 200)])
def test_one_kernel_fog_layer_equals_the_three_kernel_form_bit_for_bit(H, W, dtype, seg):
    """FogTile (k_fog_tile: f_ext, the horizontal and the vertical sums in LDS, a ring of 32 rows) against k_fog_ext / k_fog_h /
    k_fog_v: the same folds in the same order, so the same bits -- frames smaller than the 12-pixel border included (the
    reflection then bounces), segments that end inside a block of 8 rows, strips wider than the frame."""
    bg, depth = scene(H, W, 11, dtype)
    a = emu_prepass(bg, depth, 25, tiled=1, seg_rows=seg)
    b = emu_prepass(bg, depth, 25, tiled=0)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_narrow_types_are_the_float64_results_rounded_once():
    """uint8 image in, float32 fog layer and xyY map out (rr_prepass_in.in_types / rr_prepass_out.out_types): the arithmetic stays
    float64, so the narrow outputs are astype(float32) of the wide ones and the uint8 map does not move."""
    H, W = 96, 160
    bg, depth = scene(H, W, 7)
    bg8 = (bg * 255).astype(np.uint8)
    wide = emu_prepass(bg8 / 255.0, depth, 50)
    nar = emu_prepass(bg8, depth, 50, narrow=True)
    assert nar[0].dtype == np.float32 and nar[1].dtype == np.float32
    assert np.array_equal(nar[0], wide[0].astype(np.float32))
    assert np.array_equal(nar[1], wide[1].astype(np.float32))
    assert np.array_equal(nar[2], wide[2])
    f32 = emu_prepass(bg.astype(np.float32), depth, 50, narrow=True)       # a float32 image: its own values, widened
    ref = emu_prepass(bg.astype(np.float32).astype(np.float64), depth, 50)
    assert np.array_equal(f32[0], ref[0].astype(np.float32)) and np.array_equal(f32[2], ref[2])


def test_host_constants_equal_oracle():
    """The host side of the pre-pass: scalar constants and projection tables (the arithmetic itself lives on the device;
    the reference-signature calls FogRain.fog_rain_layer / generate_map are covered on the GPU tier)."""
    be, bh, num, den = fogmod.FogRain(25, 0.006, 6.0, 90, 2, 20).constants()
    obe, obh, oscale = op.fog_constants(25, 6.0, 2, 20)
    assert be == obe and bh == obh and abs(num / den - oscale) <= 1e-12 * oscale
    cw, uniq, first = envmod.EnvironmentMapGenerator(0.006, 160, 96).device_tables(96, 160)
    ocw, ouniq, ofirst = op.env_geometry(0.006, 96, 160)
    assert cw == ocw and np.array_equal(uniq, ouniq) and np.array_equal(first, ofirst)


def test_depth_samples_as_uint16_equal_the_float32_metres():
    """rr_prepass_in.depth_f64 = RR_DEPTH_U16: the uint16 samples of the depth file, metres = sample / 256 in float32
    (generator.py:366) formed where the kernels read them -- the same bits as the float32 map made by the host."""
    H, W = 96, 160
    bg, _ = scene(H, W, 9)
    rng = np.random.RandomState(4)
    d16 = (np.linspace(80, 2, H)[:, None] * np.ones((1, W)) * 256 + rng.uniform(0, 700, (H, W))).astype(np.uint16)
    a = emu_prepass(bg, d16, 50)
    b = emu_prepass(bg, d16.astype(np.float32) / np.float32(256.), 50)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    a = emu_prepass(bg, d16, 50, tiled=0)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
