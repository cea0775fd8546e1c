#!/usr/bin/env python
"""Two rocprofv3 kernel traces (csv) of the same workload run two ways (eager launches / launch plan): per kernel name the
launch count, total and average duration in each, sorted by the difference of the totals.
    python tools/trace_cmp_modes.py A_kernel_trace.csv B_kernel_trace.csv [skip_fraction]"""
import csv
import sys


def load(path, skip):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '')))
    rows.sort()
    rows = rows[int(len(rows) * skip):]
    agg = {}
    for s, e, n, q in rows:
        a = agg.setdefault(n, [0, 0])
        a[0] += 1
        a[1] += e - s
    span = rows[-1][1] - rows[0][0]
    return agg, span, len(rows)


def main():
    skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
    a, sa, na = load(sys.argv[1], skip)
    b, sb, nb = load(sys.argv[2], skip)
    print('launches %d / %d, span %.2f / %.2f ms, kernel time summed %.2f / %.2f ms' %
          (na, nb, sa / 1e6, sb / 1e6, sum(v[1] for v in a.values()) / 1e6, sum(v[1] for v in b.values()) / 1e6))
    names = sorted(set(a) | set(b), key=lambda n: -abs(b.get(n, [0, 0])[1] - a.get(n, [0, 0])[1]))
    print('%-90s %6s %9s %8s | %6s %9s %8s | %8s' % ('kernel', 'nA', 'totA ms', 'avgA us', 'nB', 'totB ms', 'avgB us', 'B-A ms'))
    for n in names[:40]:
        ca, ta = a.get(n, [0, 0])
        cb, tb = b.get(n, [0, 0])
        print('%-90s %6d %9.3f %8.1f | %6d %9.3f %8.1f | %+8.3f' %
              (n[:90], ca, ta / 1e6, ta / 1e3 / max(ca, 1), cb, tb / 1e6, tb / 1e3 / max(cb, 1), (tb - ta) / 1e6))


if __name__ == '__main__':
    main()
