"""Reads the rocprofv3 CSVs of scripts/pipe_probe.py (kernel trace + memory copy trace) and prints, batch by batch (one
k_composite per batch), when its kernels ran and how long they took in sum, next to every large copy: do upload, kernels
and download of neighbouring batches overlap, and do the kernels slow down when they do?"""
import csv
import glob
import sys

d = sys.argv[1]
K, C = [], []
for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        K.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
for f in glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        C.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Direction', ''), r))
K.sort()
C.sort()
if not K:
    sys.exit('no kernels')
t0 = min(K[0][0], C[0][0] if C else K[0][0])
ms = lambda t: (t - t0) / 1e6
# batches: split the kernel stream at every k_fog_sum (first kernel of a pipeline batch) -- or k_fov_spans without pre-pass
batches, cur = [], []
for k in K:
    name = k[2]
    if 'k_copy_pieces' in name or 'rocclr' in name:
        continue
    if ('k_fog_sum' in name or ('k_bytes_to_unit' in name)) and cur and any('k_finalize' in x[2] or 'k_png' in x[2] for x in cur):
        batches.append(cur)
        cur = []
    cur.append(k)
if cur:
    batches.append(cur)
print('%d batches' % len(batches))
for b in batches[-8:]:
    busy = sum(e - s for s, e, _ in b) / 1e6
    top = sorted(((e - s) / 1e6, n.split('(')[0][-28:]) for s, e, n in b)[-3:]
    print('  kernels %8.2f -> %8.2f ms   span %6.2f  sum of kernels %6.2f   top: %s' % (ms(b[0][0]), ms(b[-1][1]), (b[-1][1] - b[0][0]) / 1e6, busy,
                                                                                   ', '.join('%s %.2f' % (n, t) for t, n in top)))
big = [c for c in C if c[1] - c[0] > 200000]
print('%d copies longer than 0.2 ms (of %d); the last 24:' % (len(big), len(C)))
for s, e, dirn, r in big[-24:]:
    nbytes = r.get('Bytes') or r.get('Size') or ''
    print('  %-28s %8.2f -> %8.2f ms  (%5.2f ms) %s' % (dirn[:28], ms(s), ms(e), (e - s) / 1e6, nbytes))
