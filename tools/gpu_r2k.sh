#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r2k}
mkdir -p $OUT
echo "== tests"; timeout 900 python -m pytest tests/test_ops.py tests/test_trajectory.py tests/test_headline_parity.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -6 | tee $OUT/pytest.txt
B="--no-cpu-baseline --no-kernel-profile"
for i in 1 2 3; do
  echo "== 256sq off"; CONVNET_AMD_OPTIONS="igemm_256sq=0" timeout 300 python bench.py --steps 30 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-200 | tee -a $OUT/ab_off.txt
  echo "== 256sq heuristic"; timeout 300 python bench.py --steps 30 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-200 | tee -a $OUT/ab_on.txt
done
echo "== min_nkt 8"; CONVNET_AMD_OPTIONS="igemm_256sq_min_nkt=8" timeout 300 python bench.py --steps 30 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-200 | tee -a $OUT/ab_nkt8.txt
echo "== min_tiles 96"; CONVNET_AMD_OPTIONS="igemm_256sq_min_tiles=96" timeout 300 python bench.py --steps 30 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-200 | tee -a $OUT/ab_tiles96.txt
echo "== done"; date
