#!/usr/bin/env python
"""Kernels around the step boundary of a rocprofv3 kernel trace: everything that starts within +-W us of the end of the
optimizer kernel of the last-but-one step, per queue, with start / end relative to that instant.
    python tools/trace_boundary.py kernel_trace.csv [W=400]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
W = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sgd = [r for r in rows if r['Kernel_Name'].startswith('sgd_kernel')]
ref = sgd[-4] if len(sgd) >= 4 else sgd[-1]
t0 = int(ref['End_Timestamp'])
for r in rows:
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    if -W <= s <= W:
        name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ', '')[:60]
        print('q%-3s %9.1f %9.1f  %7.1f us  %s' % (r['Queue_Id'], s, e, e - s, name))
# every step boundary: idle time on the optimizer's queue between its last sgd_kernel and the next kernel there
print('all boundaries (us of idle queue behind the last sgd_kernel of a step):')
byq = [r for r in rows if r['Queue_Id'] == ref['Queue_Id']]
gaps = []
for i, r in enumerate(byq[:-1]):
    if r['Kernel_Name'].startswith('sgd_kernel') and not byq[i + 1]['Kernel_Name'].startswith('sgd_kernel'):
        gaps.append((int(byq[i + 1]['Start_Timestamp']) - int(r['End_Timestamp'])) / 1e3)
print(' '.join('%.0f' % g for g in gaps))
