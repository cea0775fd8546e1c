#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r2r}
mkdir -p $OUT
echo "== tests"; timeout 900 python -m pytest tests/test_ops.py -m gpu -q -k "bn_backward or dgrad" -p no:cacheprovider 2>&1 | tail -2
B="--no-cpu-baseline --no-kernel-profile"
run() { echo "== $1"; CONVNET_AMD_OPTIONS="$2" timeout 300 python bench.py --steps 30 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-170 | tee -a $OUT/ab_$1.txt; }
for i in 1 2 3; do
  run base ""
  run epi64 "igemm_epi_bm64=1"
done
echo "== done"; date
