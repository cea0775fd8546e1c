#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r2n}
mkdir -p $OUT
echo "== tests"; timeout 900 python -m pytest tests/test_ops.py tests/test_trajectory.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -3 | tee $OUT/pytest.txt
B="--no-cpu-baseline --no-kernel-profile"
for i in 1 2 3; do
  echo "== wgrad256 off"; CONVNET_AMD_OPTIONS="wgrad_256sq=0" timeout 300 python bench.py --steps 30 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-200 | tee -a $OUT/ab_off.txt
  echo "== wgrad256 heuristic"; timeout 300 python bench.py --steps 30 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-200 | tee -a $OUT/ab_on.txt
done
echo "== done"; date
