"""CPU tier: the library's PNG codec (readers with pooled buffers and word-wise loads / stores, the inflate with paired
literals, the deflate with vectorised run detection) compiled with AddressSanitizer + UndefinedBehaviorSanitizer and fed
damaged files, damaged streams and inputs of awkward lengths (tests/sanitize/codec_fuzz.cpp).  Skipped where the
sanitizer run-times are not installed."""
import os
import subprocess

import numpy as np
import pytest

import helpers as h


def test_codec_under_sanitizers(tmp_path):
    from PIL import Image
    src = os.path.join(h.ROOT, 'tests', 'sanitize', 'codec_fuzz.cpp')
    exe = str(tmp_path / 'codec_fuzz')
    cc = ['g++', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-std=c++17', '-ffp-contract=off',
          '-I' + os.path.join(h.ROOT, 'include'), '-I' + os.path.join(h.ROOT, 'rain-rendering_amd', 'csrc'), src, '-lz', '-lpthread', '-o', exe]
    r = subprocess.run(cc, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no sanitizer build here: " + r.stderr[-300:])
    rng = np.random.RandomState(1)
    smooth = np.clip(np.cumsum(rng.randint(-3, 4, (53, 71, 3)), axis=1) + 120, 0, 255).astype(np.uint8)
    files = []
    for name, im in (('rgb', Image.fromarray(smooth)), ('rgba', Image.fromarray(np.dstack([smooth, smooth[..., 0]]), 'RGBA')),
                     ('gray16', Image.fromarray((rng.rand(41, 59) * 65535).astype(np.uint16))),
                     ('pal', Image.fromarray(smooth).convert('P', palette=Image.ADAPTIVE, colors=9))):
        p = str(tmp_path / (name + '.png'))
        im.save(p)
        files.append(p)
    run = subprocess.run([exe] + files, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS='detect_leaks=0'))          # (the buffer pools live until exit by design)
    assert run.returncode == 0, (run.stdout[-600:], run.stderr[-3000:])
    assert 'wrong 0' in run.stdout and 'readers:' in run.stdout and 'device payloads: wrong 0' in run.stdout


def test_device_byte_code_under_sanitizers(tmp_path):
    """The host build of the device's entropy coder (csrc/rr_deflate.h) and scanline-filter rule (csrc/rr_pngrows.h) with
    AddressSanitizer + UndefinedBehaviorSanitizer on inputs of every awkward length in exact-size buffers
    (tests/sanitize/device_code_fuzz.cpp): zlib inflates every stream back."""
    src = os.path.join(h.ROOT, 'tests', 'sanitize', 'device_code_fuzz.cpp')
    exe = str(tmp_path / 'device_code_fuzz')
    cc = ['g++', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-std=c++17', '-ffp-contract=off',
          '-I' + os.path.join(h.ROOT, 'include'), '-I' + os.path.join(h.ROOT, 'rain-rendering_amd', 'csrc'), src, '-lz', '-o', exe]
    r = subprocess.run(cc, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no sanitizer build here: " + r.stderr[-300:])
    run = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=dict(os.environ, ASAN_OPTIONS='detect_leaks=0'))
    assert run.returncode == 0, (run.stdout[-600:], run.stderr[-3000:])
    assert 'wrong 0' in run.stdout and 'unfilter: wrong 0' in run.stdout and 'coded' in run.stdout
