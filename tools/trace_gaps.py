"""Idle time in a rocprofv3 kernel trace (…_kernel_trace.csv) of bench.py: for the last complete step, the union of
all kernel intervals, and per HSA queue the gaps between consecutive kernels grouped by (previous kernel -> next)."""
import csv, re, sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'nchw_to_pairs' in r['Kernel_Name'] or 'nchw_to_nhwc' in r['Kernel_Name']]
a, b = marks[-2], marks[-1]
step = rows[a:b]
t0, t1 = int(step[0]['Start_Timestamp']), int(rows[b]['Start_Timestamp'])


def short(n):
    n = re.sub(r'\(.*', '', n).replace('void ', '')
    return re.sub(r'<.*', '', n)


busy, end = 0, t0
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if e > end:
        busy += e - max(s, end)
        end = e
print('step %.1f us, union of kernels %.1f us (%.1f %% busy), %d kernels' % ((t1 - t0) / 1e3, busy / 1e3, 100.0 * busy / (t1 - t0), len(step)))
gaps = defaultdict(lambda: [0, 0.0])
last = {}
for r in step:
    q = r['Queue_Id']
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if q in last:
        g = (s - last[q][0]) / 1e3
        if 0 < g < 200:
            k = (short(last[q][1]), short(r['Kernel_Name']))
            gaps[k][0] += 1
            gaps[k][1] += g
    last[q] = (e, r['Kernel_Name'])
tot = sum(v[1] for v in gaps.values())
print('same-queue gaps (0 < g < 200 us): %.1f us in total' % tot)
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print('  %-28s -> %-28s n=%3d  avg %5.1f us  total %6.1f us' % (k[0][:28], k[1][:28], v[0], v[1] / v[0], v[1]))
# the dependent chain (everything but the weight-gradient kernels of the side stream), whatever queue it ran on
chain = [r for r in step if not short(r['Kernel_Name']).startswith('wgrad_')]
cg = defaultdict(lambda: [0, 0.0])
end, prev = None, None
qhops = 0
for r in chain:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if end is not None and s > end:
        k = (short(prev['Kernel_Name']), short(r['Kernel_Name']), 'hop' if prev['Queue_Id'] != r['Queue_Id'] else 'same')
        cg[k][0] += 1
        cg[k][1] += (s - end) / 1e3
    if prev is not None and prev['Queue_Id'] != r['Queue_Id']:
        qhops += 1
    if end is None or e > end:
        end, prev = e, r
print('dependent chain (no wgrad_*): %d kernels, %d queue changes, idle between them %.1f us' % (len(chain), qhops, sum(v[1] for v in cg.values())))
for k, v in sorted(cg.items(), key=lambda kv: -kv[1][1])[:12]:
    print('  %-26s -> %-26s %-4s n=%3d  avg %5.1f us  total %6.1f us' % (k[0][:26], k[1][:26], k[2], v[0], v[1] / v[0], v[1]))
dur = defaultdict(lambda: [0, 0.0])
for r in step:
    d = dur[short(r['Kernel_Name'])]
    d[0] += 1
    d[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print('small kernels (avg < 15 us):')
for k, v in sorted(dur.items(), key=lambda kv: -kv[1][1]):
    if v[1] / v[0] < 15:
        print('  %-34s n=%3d avg %5.1f us total %6.1f us' % (k[:34], v[0], v[1] / v[0], v[1]))
