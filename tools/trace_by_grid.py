"""Group a rocprofv3 kernel_trace.csv by (kernel, grid size): per-shape average duration, so the
layers behind one kernel name can be told apart (measurement aid)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
d = collections.OrderedDict()
for r in rows:
    name = r['Kernel_Name']
    short = name.split('(')[0][-70:]
    key = (short, r.get('Grid_Size', r.get('Grid_Size_X', '?')))
    d.setdefault(key, []).append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = sum(sum(v) for v in d.values())
out = sorted(d.items(), key=lambda kv: -sum(kv[1]))
for (name, grid), v in out[:60]:
    print('%-72s grid %9s  n=%4d  avg %8.1f us  total %8.2f ms  %5.1f%%' % (name, grid, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6, 100.0 * sum(v) / tot))
