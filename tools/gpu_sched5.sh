#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-sched5}
mkdir -p $OUT
run() { name=$1; shift; echo -n "$name: "; env "$@" timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-profile 2>&1 | grep '"metric"\|Error\|error' | head -2 | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(d['value'], d['ms_per_step'])
except Exception: print('FAILED', t[:300])"; }
{
run base X=1
run epi_bm64 CONVNET_AMD_OPTIONS=igemm_epi_bm64=1
run junc_min60 CONVNET_AMD_FUSE_BN_BWD_JUNC_MIN_MB=60
run junc_min120 CONVNET_AMD_FUSE_BN_BWD_JUNC_MIN_MB=120
run junc_min250 CONVNET_AMD_FUSE_BN_BWD_JUNC_MIN_MB=250
run base2 X=1
run wg3_192 CONVNET_AMD_OPTIONS=wgrad_3x3_wgs=192
run wg3_512 CONVNET_AMD_OPTIONS=wgrad_3x3_wgs=512
run lazy300 CONVNET_AMD_LAZY_DY_MIN_MB=300
run lazy90 CONVNET_AMD_LAZY_DY_MIN_MB=90
run base3 X=1
} 2>&1 | tee $OUT/sched.txt
