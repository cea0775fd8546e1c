#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) kernel trace into the `--stats`-style table
(name, calls, total ns, average ns, %), optionally skipping the first N dispatches (warm-up)."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute('select name, duration, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, lds_size '
                      'from kernels order by start').fetchall()
    agg = {}
    for name, dur, gx, wx, v, a, lds in rows:
        e = agg.setdefault(name, [0, 0, v, a, lds])
        e[0] += 1
        e[1] += dur
    tot = sum(e[1] for e in agg.values())
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","VGPR","AGPR","LDS"')
    for name, e in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        short = name if len(name) < 110 else name[:107] + '...'
        print('"%s",%d,%d,%.0f,%.2f,%s,%s,%s' % (short, e[0], e[1], e[1] / e[0], 100.0 * e[1] / tot, e[2], e[3], e[4]))
    print('"TOTAL",%d,%d,,100.0' % (len(rows), tot))


if __name__ == '__main__':
    main()
