#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r2o}
mkdir -p $OUT
B="--no-cpu-baseline --no-kernel-profile"
run() { echo "== $1"; CONVNET_AMD_OPTIONS="$2" timeout 300 python bench.py --steps 30 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-170 | tee -a $OUT/ab_$1.txt; }
for i in 1 2; do
  run base ""
  run nkt12 "igemm_256sq_min_nkt=12"
  run tiles128 "igemm_256sq_min_tiles=128"
  run tiles192 "igemm_256sq_min_tiles=192"
  run nofragdb "igemm_variant_256=11"
done
echo "== done"; date
