"""Registers, LDS, scratch and occupancy of every kernel, keyed by its demangled name (hipcc -Rpass-analysis=kernel-resource-usage;
no GPU needed).  python scripts/kernel_resources.py [out.txt]   -- also leaves the gfx950 assembly in /tmp/rr_isa/ for reading."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'rain-rendering_amd', 'csrc')
tmp = '/tmp/rr_isa'
os.makedirs(tmp, exist_ok=True)
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-Wno-unused-value', '-Wno-unused-result',
       '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '--save-temps', '-Rpass-analysis=kernel-resource-usage', '-c', os.path.join(CSRC, 'rainhip.hip'),
       '-o', os.path.join(tmp, 'rainhip.o')]
r = subprocess.run(cmd, cwd=tmp, capture_output=True, text=True)
txt = r.stderr
if r.returncode != 0:
    sys.stderr.write('\n'.join(l for l in txt.split('\n') if 'error' in l or 'note:' in l)[:4000] + '\n')
    raise SystemExit('hipcc failed (%d)' % r.returncode)
rows = []
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split('\n')[0].split(' [-Rpass')[0].strip()
    g = lambda k: int(re.search(k + r': (\d+)', b).group(1)) if re.search(k + r': (\d+)', b) else -1
    rows.append((name, g('VGPRs'), g('AGPRs'), g('SGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
dem = subprocess.run(['c++filt'] + [r[0] for r in rows], capture_output=True, text=True).stdout.split('\n')
lines = ['%-44s %5s %5s %5s %8s %10s %9s' % ('kernel', 'VGPR', 'AGPR', 'SGPR', 'scratch', 'waves/SIMD', 'LDS bytes')]
for r, d in sorted(zip(rows, dem), key=lambda x: x[1]):
    d = re.sub(r'\(anonymous namespace\)::', '', d)
    d = re.sub(r'^void ', '', d)
    d = re.sub(r'\(.*', '', d)
    lines.append('%-44s %5d %5d %5d %8d %10d %9d' % ((d[:44],) + r[1:]))
out = '\n'.join(lines) + '\n(static LDS only: dynamic shared memory is sized at launch)\n'
if len(sys.argv) > 1:
    open(sys.argv[1], 'w').write(out)
print(out)
