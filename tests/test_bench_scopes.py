"""bench.py finds a timing scope's counters (rocprofv3 --pmc passes) under the NAMES OF THE KERNELS launched inside the scope.
Round 5 renamed the dominant kernel (k_blur_fused_dma under the scope k_blur_fused) and the bench line lost its traffic figure
until the mapping followed.  This test reads the launches of every ProfScope block out of rainhip.hip and asks bench.py's
mapping for each of them."""
import importlib.util
import os
import re
import sys

import helpers as h


def _bench():
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(h.ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ['bench.py']
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_every_kernel_of_a_timing_scope_maps_to_it():
    src = open(os.path.join(h.ROOT, 'rain-rendering_amd', 'csrc', 'rainhip.hip')).read()
    bench = _bench()
    pairs = set()
    for m in re.finditer(r'ProfScope ps\(ctx, \w+, "([a-z_0-9]+)"\);', src):
        name, p = m.group(1), m.start()
        k = src.rfind('{', 0, p)                      # the block the scope object lives in, up to its closing brace
        depth = 0
        while True:
            if src[k] == '{':
                depth += 1
            elif src[k] == '}':
                depth -= 1
                if depth == 0:
                    break
            k += 1
        body = src[p:k]
        for km in re.finditer(r'hipLaunchKernelGGL\(\(?(k_[a-z_0-9]+(?:<[01]>)?)', body):
            pairs.add((name, km.group(1)))
        for km in re.finditer(r'launch(?:32)?\((k_[a-z_0-9]+)', body):
            pairs.add((name, km.group(1)))
        if 'RR_COMP32(' in body:
            pairs.add((name, 'k_composite32'))
    assert len(pairs) > 30 and ('k_blur_fused', 'k_blur_fused_dma') in pairs and ('k_blur_cols', 'k_blur<1>') in pairs, sorted(pairs)
    bad = sorted((n, k) for n, k in pairs if bench.scope_of_name('void (anonymous namespace)::' + k + '(args)') != n)
    assert not bad, bad
