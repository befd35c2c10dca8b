"""Reads the rocprofv3 CSVs of scripts/pipe_probe.py (kernel trace + memory copy trace) and prints a compact timeline: for
every kernel chain (one k_composite per batch) its span, and the copies / copy kernels that overlap it."""
import csv
import glob
import sys

d = sys.argv[1]
ev = []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K', r['Kernel_Name'][:60], r.get('Queue_Id', ''), r.get('Stream_Id', '')))
for f in glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C', r.get('Direction', r.get('Name', '')), '', r.get('Stream_Id', '')))
ev.sort()
if not ev:
    sys.exit('no events')
t0 = ev[0][0]
# merge into lanes: copy kernels, memcpy H2D, memcpy D2H, compute kernels
def lane(e):
    if e[2] == 'C':
        return 'memcpy ' + e[3][:24]
    if 'k_copy_pieces' in e[3]:
        return 'k_copy_pieces q' + e[4]
    return 'compute'
spans = {}
for e in ev:
    spans.setdefault(lane(e), []).append((e[0] - t0, e[1] - t0))
def merged(iv, gap=20000):
    out = []
    for a, b in sorted(iv):
        if out and a - out[-1][1] < gap:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out
tail = ev[-1][1] - t0
print('total %.1f ms' % (tail / 1e6))
for k, iv in spans.items():
    m = merged(iv)
    busy = sum(b - a for a, b in m)
    print('%-40s busy %7.1f ms in %4d runs; last 12 runs (ms): %s' % (k, busy / 1e6, len(m), ' '.join('%.1f-%.1f' % (a / 1e6, b / 1e6) for a, b in m[-12:])))
