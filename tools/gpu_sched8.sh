#!/bin/bash
# igemm_8w default: interleaved whole-step A/B (16 vs 0), five pairs on one box
export TMPDIR=/tmp
OUT=gpurun_out/${1:-sched8}
mkdir -p $OUT
run() { name=$1; shift; echo -n "$name: "; env "$@" timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-profile 2>&1 | grep '"metric"\|Error\|error' | head -2 | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(d['value'], d['ms_per_step'], d['config']['final_loss'])
except Exception: print('FAILED', t[:300])"; }
{
run warm X=1
for i in 1 2 3 4 5; do
run w16_$i X=1
run w0_$i CONVNET_AMD_OPTIONS=igemm_8w=0
done
} 2>&1 | tee $OUT/sched.txt
