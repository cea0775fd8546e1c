#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3l}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops.py tests/test_trajectory.py -x -q -m gpu -k "conv3x3_halo or full_size" 2>&1 | tail -5 | tee $OUT/pytest.txt
run() { name=$1; shift; echo -n "$name: "; env "$@" timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-profile 2>&1 | grep '"metric"\|Error\|error' | head -2 | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(d['value'], d['ms_per_step'], d['config']['final_loss'])
except Exception: print('FAILED', t[:300])"; }
{
run warm X=1
for i in 1 2 3; do
run halo_$i X=1
run nohalo_$i CONVNET_AMD_CONV3X3_HALO=0
run halo1024_$i CONVNET_AMD_OPTIONS=conv3x3_wgs=1024
done
} 2>&1 | tee $OUT/sched.txt
bash tools/gpu_r3g.sh ${1:-r3l}/prof 2>&1 | grep -E "conv3x3|^[0-9]|total"
