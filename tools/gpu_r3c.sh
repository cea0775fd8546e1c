#!/bin/bash
# per-kernel (alone) timings of the junction forward with and without lazy z
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3c}
mkdir -p $OUT
for lz in 0 1; do
CONVNET_AMD_LAZY_Z=$lz timeout 600 python bench.py --steps 10 --warmup 5 --no-cpu-baseline > $OUT/bench_lz$lz.json 2> $OUT/bench_lz$lz.err
python - <<P
import json
d=json.loads([l for l in open('$OUT/bench_lz$lz.json') if l.startswith('{')][-1])
print('LAZY_Z=$lz', d['value'], d['ms_per_step'])
for name in ('kernels',):
    for k,v in sorted(d[name].items(), key=lambda kv:-kv[1]['ms_per_step']):
        if 'bn_' in k or 'lazy z' in k or '1, 4, 2, 1, 1, false, false, false, false, false' in k or '2, 4, 2, 1, 1, false, false, false, false, false' in k:
            print('  %-100s %7.3f ms n=%3d avg %7.1f us %6.0f GB/s' % (k[:100], v['ms_per_step'], v['launches_per_step'], v['avg_us_per_launch'], v.get('gbs') or 0))
for k,v in d['conv_layers'].items() if isinstance(d.get('conv_layers'),dict) else []:
    pass
P
done 2>&1 | tee $OUT/summary.txt
