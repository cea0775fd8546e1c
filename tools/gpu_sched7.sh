#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-sched7}
mkdir -p $OUT
run() { name=$1; shift; echo -n "$name: "; env "$@" timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-profile 2>&1 | grep '"metric"\|Error\|error' | head -2 | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(d['value'], d['ms_per_step'], d['config']['final_loss'])
except Exception: print('FAILED', t[:300])"; }
{
run warm X=1
run new X=1
run old8w0 CONVNET_AMD_OPTIONS=igemm_8w=0
run new2 X=1
run old_all CONVNET_AMD_OPTIONS=igemm_8w=0,igemm_epi_8w=0
run new3 X=1
run r02like CONVNET_AMD_OPTIONS=igemm_8w=0,igemm_epi_8w=0,igemm_ilv=0,wgrad_3x3=0,igemm_dma_min_nkt=24 CONVNET_AMD_LAZY_DY=0
} 2>&1 | tee $OUT/sched.txt
