#!/usr/bin/env python
"""rocprofv3 FETCH_SIZE / WRITE_SIZE passes -> per-kernel HBM traffic per launch (JSON).

Units and corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are
in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of wide coalesced reads, so it is
doubled (calibrated here on bn_stats_kernel, a pure streaming read of a known tensor size);
WRITE_SIZE matched the known store volume of bn_apply_kernel and is used as is."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    agg, calls = defaultdict(float), defaultdict(set)
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            agg[r['Kernel_Name']] += float(r['Counter_Value'])
            calls[r['Kernel_Name']].add(r['Dispatch_Id'])
    return agg, {k: len(v) for k, v in calls.items()}


def collect(root, steps=3):
    """{source, kernels: {name: per-launch bytes}, profiled_steps, hbm_bytes_per_step} from <root>/fetch and
    <root>/write (one rocprofv3 --pmc pass each)."""
    fetch, fc = load(os.path.join(root, 'fetch'), 'FETCH_SIZE')
    write, wc = load(os.path.join(root, 'write'), 'WRITE_SIZE')
    res = {}
    for k in sorted(set(fetch) | set(write)):
        n = max(fc.get(k, 0), wc.get(k, 0), 1)
        rd = 2.0 * fetch.get(k, 0.0) * 1024.0
        wr = write.get(k, 0.0) * 1024.0
        res[k] = {'launches': n, 'read_bytes_per_launch': rd / n, 'write_bytes_per_launch': wr / n,
                  'hbm_bytes_per_launch': (rd + wr) / n}
    total = sum(v['hbm_bytes_per_launch'] * v['launches'] for v in res.values())
    return {'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950)',
            'kernels': res, 'profiled_steps': steps, 'hbm_bytes_per_step': total / steps,
            'workload': 'bench.py defaults (ResNet-50 bf16 b=256, 1 GPU)'}


def main():
    root, out = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3    # tools/pmc_round.sh profiles bench.py --steps 2 --warmup 1
    doc = collect(root, steps)
    res, total = doc['kernels'], doc['hbm_bytes_per_step'] * steps
    json.dump(doc, open(out, 'w'), indent=1, sort_keys=True)
    print('all kernels: %.1f GB of HBM traffic per step (%d profiled steps)' % (total / steps / 1e9, steps))
    top = sorted(res.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:12]
    for k, v in top:
        print('%-90s n=%4d  %.1f MB/launch' % (k[:90], v['launches'], v['hbm_bytes_per_launch'] / 1e6))


if __name__ == '__main__':
    main()
