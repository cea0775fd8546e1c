#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-sched2}
mkdir -p $OUT
run() { name=$1; shift; echo -n "$name: "; env "$@" timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-profile 2>/dev/null | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
{
run base X=1
for v in 128 192 256 320 384; do run wg3_$v CONVNET_AMD_OPTIONS=wgrad_3x3_wgs=$v; done
for v in 256 384 768; do run wgt_$v CONVNET_AMD_OPTIONS=wgrad_target_wgs=$v; done
run wg3_256_wgt_384 CONVNET_AMD_OPTIONS=wgrad_3x3_wgs=256,wgrad_target_wgs=384
run wg3_256_wgt_256 CONVNET_AMD_OPTIONS=wgrad_3x3_wgs=256,wgrad_target_wgs=256
for v in 1024 4096; do run bnapply_$v CONVNET_AMD_OPTIONS=bn_apply_blocks=$v; done
for v in 256 1024; do run bnreduce_$v CONVNET_AMD_OPTIONS=bn_reduce_blocks=$v; done
run base2 X=1
} 2>&1 | tee $OUT/sched.txt
