"""CPU tier: the kernel arithmetic (rr_device.h compiled with g++, tests/hostemu) against
the numpy oracle.  This is what lets arithmetic bugs surface without a GPU; the GPU tier
repeats the same comparisons through the real kernels."""
import numpy as np
import pytest

import helpers as h


@pytest.mark.parametrize("H,W,N,seed,noise", [(96, 160, 150, 10, 0.0), (128, 256, 200, 20, 3.0), (64, 64, 80, 30, 0.0)])
def test_hostemu_matches_oracle(tmp_path, H, W, N, seed, noise):
    sc = h.Scene(tmp_path, H, W, N, seed0=seed)
    bg, env = sc.frame_inputs(0)
    drops = sc.product_drops(0, noise_std=noise, noise_scale=1.0)
    emu = h.emu_render(sc, bg, bg, env, drops)
    ref = h.oracle_render(sc, 0, bg, bg, env, faithful=True, noise_std=noise, noise_scale=1.0)
    assert np.array_equal(emu['status'], ref['status'])
    assert np.array_equal(emu['mask'], ref['mask'])
    assert np.array_equal(emu['mask_i32'], ref['mask_i32'])
    assert np.abs(emu['image_u8'].astype(int) - ref['image_u8'].astype(int)).max() <= 1
    assert np.abs(emu['rainy_bg'] - ref['rainy_bg']).max() < 1e-12
