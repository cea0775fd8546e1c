#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3w}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops.py tests/test_trajectory.py -x -q -m gpu -k "lazy_a or streaming_conv1x1" 2>&1 | tail -5 | tee $OUT/pytest.txt
run() { name=$1; shift; echo -n "$name: "; env "$@" timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-profile 2>&1 | grep '"metric"\|Error\|error' | head -2 | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(d['value'], d['ms_per_step'], d['config']['final_loss'])
except Exception: print('FAILED', t[:300])"; }
{
run warm X=1
for i in 1 2 3; do
run lazy_a_$i X=1
run off_$i CONVNET_AMD_LAZY_A=0
done
} 2>&1 | tee $OUT/sched.txt
