"""Which kernels surround every launch of NAME in a rocprofv3 kernel_trace.csv (same queue, by start time)?  Measurement aid.
    python tools/trace_neighbors.py kernel_trace.csv __amd_rocclr_copyBuffer"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
name = sys.argv[2]
kn = 'Kernel_Name' if 'Kernel_Name' in rows[0] else 'Name'
rows.sort(key=lambda r: int(r['Start_Timestamp']))
byq = collections.defaultdict(list)
for r in rows:
    byq[r.get('Queue_Id', '0')].append(r)
pairs = collections.Counter()
for q, rs in byq.items():
    for i, r in enumerate(rs):
        if name in r[kn]:
            prev = rs[i - 1][kn][:60] if i else '-'
            nxt = rs[i + 1][kn][:60] if i + 1 < len(rs) else '-'
            pairs[(q, prev, nxt)] += 1
for (q, p, n), c in pairs.most_common(25):
    print('%4d  queue %s  after %-62s before %s' % (c, q, p, n))
print('total', sum(pairs.values()), 'queues', {q: len(rs) for q, rs in byq.items()})
