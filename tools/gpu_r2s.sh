#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r2s}
mkdir -p $OUT
echo "== gpu tests"; timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -12 | tee $OUT/pytest_gpu.txt
B="--no-cpu-baseline --no-kernel-profile"
echo "== bench bf16"; timeout 300 python bench.py --steps 30 --warmup 5 $B 2>&1 | grep '"metric"' | tee $OUT/bench_bf16.txt | cut -c80-200
echo "== bench f16"; timeout 300 python bench.py --dtype f16 --steps 30 --warmup 5 $B 2>&1 | grep '"metric"' | tee $OUT/bench_f16.txt | cut -c80-200
echo "== done"; date
