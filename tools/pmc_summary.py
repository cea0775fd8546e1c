#!/usr/bin/env python
"""Aggregate rocprofv3 counter_collection CSVs (one per PMC pass) per kernel name."""
import csv
import glob
import os
import sys
from collections import defaultdict


def load(path):
    rows = []
    for f in glob.glob(os.path.join(path, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    return rows


def main():
    root = sys.argv[1]
    for name in sorted(os.listdir(root)):
        d = os.path.join(root, name)
        if not os.path.isdir(d):
            continue
        rows = load(d)
        if not rows:
            continue
        agg = defaultdict(lambda: defaultdict(float))
        calls = defaultdict(set)
        for r in rows:
            k = r.get('Kernel_Name', '?')
            k = k if len(k) < 70 else k[:67] + '...'
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            calls[k].add(r.get('Dispatch_Id'))
        ctrs = sorted({c for v in agg.values() for c in v})
        print('== pass %s: per-kernel SUM over dispatches (calls)' % name)
        print('kernel,calls,' + ','.join(ctrs))
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
            print('"%s",%d,' % (k, len(calls[k])) + ','.join('%.4g' % v.get(c, 0.0) for c in ctrs))


if __name__ == '__main__':
    main()
