#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3g}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_ops.py tests/test_trajectory.py -x -q -m gpu -k "junction_pair" 2>&1 | tail -3 | tee $OUT/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - <<P
import json
d=json.loads([l for l in open('$OUT/bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'])
for name in ('kernels_overlapped','kernels'):
    print('==',name)
    tot=0
    for k,v in sorted(d[name].items(), key=lambda kv:-kv[1]['ms_per_step']):
        tot+=v['ms_per_step']
        if v['ms_per_step']>=0.08: print('  %-112s %7.3f ms n=%3d avg %7.1f us %6.0f GB/s' % (k[:112], v['ms_per_step'], v['launches_per_step'], v['avg_us_per_launch'], v.get('gbs') or 0))
    print('  total', tot)
P
