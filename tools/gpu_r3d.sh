#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3d}
mkdir -p $OUT
bash tools/gpu_r3c.sh ${1:-r3d}/prof
timeout 600 python -m pytest tests/test_ops.py tests/test_trajectory.py -x -q -m gpu -k "lazy_z" 2>&1 | tail -3 | tee $OUT/pytest.txt
run() { name=$1; shift; echo -n "$name: "; env "$@" timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-profile 2>&1 | grep '"metric"\|Error\|error' | head -2 | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(d['value'], d['ms_per_step'], d['config']['final_loss'])
except Exception: print('FAILED', t[:300])"; }
{
run warm X=1
for i in 1 2 3; do
run nolz_$i CONVNET_AMD_LAZY_Z=0
run lz_$i CONVNET_AMD_LAZY_Z=1
done
} 2>&1 | tee $OUT/sched.txt
