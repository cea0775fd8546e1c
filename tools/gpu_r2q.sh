#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r2q}
mkdir -p $OUT
echo "== test"; timeout 600 python -m pytest tests/test_ops.py -m gpu -q -k "persistent" -p no:cacheprovider 2>&1 | tail -2
echo "== layers pers=0 (variant 1 forced)"; timeout 600 python tools/bench_layers.py --variants 1 --iters 10 2>&1 | tail -25 | cut -c1-72,100-130 | tee $OUT/layers_pers0.txt
echo "== layers pers=1"; CONVNET_AMD_OPTIONS="igemm_pers=1" timeout 600 python tools/bench_layers.py --variants 1 --iters 10 2>&1 | tail -25 | cut -c1-72,100-130 | tee $OUT/layers_pers1.txt
B="--no-cpu-baseline --no-kernel-profile"
for i in 1 2; do
  echo "== step pers off"; timeout 300 python bench.py --steps 30 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-170 | tee -a $OUT/ab_off.txt
  echo "== step pers on"; CONVNET_AMD_OPTIONS="igemm_pers=1" timeout 300 python bench.py --steps 30 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-170 | tee -a $OUT/ab_on.txt
done
echo "== done"; date
