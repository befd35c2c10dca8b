"""CPU tier: the C-ABI library loads, exports every symbol include/rainhip.h declares, its
struct layouts match the ctypes/numpy mirrors, and the product path FAILS LOUDLY without a
GPU (no CPU fallback anywhere behind the ABI)."""
import ctypes
import os
import re

import numpy as np
import pytest

import helpers as h


def _declared():
    hdr = open(os.path.join(h.ROOT, 'include', 'rainhip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    return sorted(set(re.findall(r'\b(rr_[a-z0-9_]+)\s*\(', hdr)))


def test_library_exports_every_declared_symbol(built):
    lib = h.hb.load_library()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "librainhip.so does not export %s" % n
    assert sorted(h.hb.EXPORTS) == names
    assert lib.rr_version() == 400


def test_struct_layouts(built):
    lib = h.hb.load_library()
    assert lib.rr_sizeof_drop() == h.hb.DROP_DTYPE.itemsize == 112
    assert lib.rr_sizeof_camera() == ctypes.sizeof(h.hb.rr_camera)
    assert lib.rr_sizeof_frame_in() == ctypes.sizeof(h.hb.rr_frame_in)
    assert lib.rr_sizeof_frame_out() == ctypes.sizeof(h.hb.rr_frame_out)
    offs = {k: v[1] for k, v in h.hb.DROP_DTYPE.fields.items()}
    assert offs['x0'] == 0 and offs['iw1'] == 32 and offs['wps'] == 48 and offs['wpe'] == 72 and offs['rot_cos'] == 96


def test_no_gpu_means_loud_failure(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    hnd = ctypes.c_void_p()
    assert h.hb.load_library().rr_create(ctypes.byref(hnd), 0) == -2      # RR_E_NO_DEVICE
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        h.hb.RainHip(0)


def test_missing_library_is_loud(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        h.hb.load_library(str(tmp_path / 'nope.so'))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: no file of the product package may mention it."""
    pkg = os.path.join(h.ROOT, 'rain-rendering_amd')
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
                assert 'hostemu' not in src or f in ('rr_device.h', 'rr_prepass.h', 'rr_particles.h', 'rr_deflate.h', 'rr_pngrows.h'), f


def test_library_reads_no_environment_switches():
    """Debug / A-B switches live behind rr_set_option (never changing a result bit), not in environment variables:
    a stray variable must not be able to produce a fast wrong answer."""
    for f in ('rainhip.hip', 'rr_host.cpp', 'rr_device.h', 'rr_prepass.h'):
        src = open(os.path.join(h.ROOT, 'rain-rendering_amd', 'csrc', f)).read()
        assert 'getenv' not in src, f
