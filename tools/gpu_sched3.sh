#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-sched3}
mkdir -p $OUT
run() { name=$1; shift; echo -n "$name: "; env "$@" timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-profile 2>&1 | grep '"metric"\|Error\|error' | head -2 | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(d['value'], d['ms_per_step'])
except Exception: print('FAILED', t[:300])"; }
{
run base X=1
for v in 2 3 4 5 6 7; do run cus_$v CONVNET_AMD_WGRAD_CUS=$v; done
run cus_4_wgt768 CONVNET_AMD_WGRAD_CUS=4 CONVNET_AMD_OPTIONS=wgrad_target_wgs=768
run bnreduce_256 CONVNET_AMD_OPTIONS=bn_reduce_blocks=256
run bnreduce_384 CONVNET_AMD_OPTIONS=bn_reduce_blocks=384
run base2 X=1
} 2>&1 | tee $OUT/sched.txt
