#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r2g}
mkdir -p $OUT
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -2 | tee $OUT/bench.txt | cut -c1-300
echo "== bench quantize bf16 b=256"; timeout 600 python bench.py --quantize --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -2 | tee $OUT/bench_quant_bf16.txt | cut -c1-300
echo "== done"; date
