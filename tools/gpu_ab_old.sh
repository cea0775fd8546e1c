#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-abold}
mkdir -p $OUT
B="--no-cpu-baseline --no-kernel-profile"
for i in 1 2 3; do
  echo "== new"; timeout 300 python bench.py --steps 30 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-170 | tee -a $OUT/ab_new.txt
  echo "== old (bb181e0, before fp16 / XF)"; (cd _ab_old && timeout 300 python bench.py --steps 30 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-170) | tee -a $OUT/ab_old.txt
done
echo "== done"
