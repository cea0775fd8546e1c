#!/usr/bin/env python
"""Per-step timeline of two rocprofv3 kernel traces of the same step (eager / plan): for the k-th launch of selected kernels
within a step, its start offset from the step's first kernel and its duration, averaged over the steady-state steps; plus
per-queue busy time and the overlap of the two busiest queues per step.
    python tools/trace_step_timeline.py A.csv B.csv [kernel-substring ...]"""
import csv
import sys
from collections import defaultdict

FIRST = 'nchw_to_pairs_kernel'


def load(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0')))
    rows.sort()
    steps, cur = [], None
    for r in rows:
        if r[2].startswith(FIRST):
            cur = []
            steps.append(cur)
        if cur is not None:
            cur.append(r)
    return steps[len(steps) // 2:-1]      # steady state, the last (possibly cut) step dropped


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    out = []
    for s, e in iv:
        if cs is None or s > ce:
            if cs is not None:
                out.append((cs, ce))
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if cs is not None:
        out.append((cs, ce))
    return out


def inter(a, b):
    i = j = 0
    t = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if e > s:
            t += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return t


def summarise(steps, pats):
    n = len(steps)
    per = defaultdict(lambda: [0.0, 0.0, 0])
    qbusy = defaultdict(float)
    ovl = 0.0
    span = 0.0
    for st in steps:
        t0 = st[0][0]
        span += (max(r[1] for r in st) - t0) / 1e3
        cnt = defaultdict(int)
        byq = defaultdict(list)
        for s, e, name, q in st:
            byq[q].append((s, e))
            for p in pats:
                if p in name:
                    k = cnt[p]
                    cnt[p] += 1
                    a = per[(p, k)]
                    a[0] += (s - t0) / 1e3
                    a[1] += (e - s) / 1e3
                    a[2] += 1
        us = {q: union(v) for q, v in byq.items()}
        for q, u in us.items():
            qbusy[q] += sum(e - s for s, e in u) / 1e3
        qs = sorted(us, key=lambda q: -sum(e - s for s, e in us[q]))[:2]
        if len(qs) == 2:
            ovl += inter(us[qs[0]], us[qs[1]]) / 1e3
    return n, span / n, {q: v / n for q, v in qbusy.items()}, ovl / n, {k: (v[0] / v[2], v[1] / v[2]) for k, v in per.items()}


def main():
    pats = sys.argv[3:] or ['wgrad3x3_kernel<bf16_t, 128>', 'jdgrad_w32_kernel']
    A = summarise(load(sys.argv[1]), pats)
    B = summarise(load(sys.argv[2]), pats)
    for tag, S in (('A', A), ('B', B)):
        print('%s: %d steps, span %.1f us, queue busy %s, overlap of the two busiest queues %.1f us' %
              (tag, S[0], S[1], {q: round(v, 1) for q, v in S[2].items()}, S[3]))
    print('%-40s %3s | %10s %8s | %10s %8s | %8s %8s' % ('kernel', 'k', 'startA us', 'durA', 'startB us', 'durB', 'dstart', 'ddur'))
    for key in sorted(A[4], key=lambda k: (k[0], k[1])):
        if key in B[4]:
            a, b = A[4][key], B[4][key]
            print('%-40s %3d | %10.1f %8.1f | %10.1f %8.1f | %+8.1f %+8.1f' % (key[0][:40], key[1], a[0], a[1], b[0], b[1], b[0] - a[0], b[1] - a[1]))


if __name__ == '__main__':
    main()
