#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-sched4}
mkdir -p $OUT
run() { name=$1; shift; echo -n "$name: "; env "$@" timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-profile 2>&1 | grep '"metric"\|Error\|error' | head -2 | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(d['value'], d['ms_per_step'])
except Exception: print('FAILED', t[:300])"; }
{
run base X=1
run inner1x1 CONVNET_AMD_FUSE_BN_BWD_INNER_1X1=1
run inner_mb30 CONVNET_AMD_FUSE_BN_BWD_INNER_MB=30
run inner_mb60 CONVNET_AMD_FUSE_BN_BWD_INNER_MB=60
run inner_mb120 CONVNET_AMD_FUSE_BN_BWD_INNER_MB=120
run inner1x1_mb60 CONVNET_AMD_FUSE_BN_BWD_INNER_1X1=1 CONVNET_AMD_FUSE_BN_BWD_INNER_MB=60
run base2 X=1
run graph CONVNET_AMD_GRAPH=1
run marks0 CONVNET_AMD_MARKS=0
run base3 X=1
} 2>&1 | tee $OUT/sched.txt
