#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r2f}
mkdir -p $OUT
echo "== quant + abi tests"; timeout 900 python -m pytest tests/test_quant.py tests/test_abi.py -q --tb=short -p no:cacheprovider 2>&1 | tail -30 | tee $OUT/pytest_quant.txt
echo "== bench quantize bf16 b=256"; timeout 600 python bench.py --quantize --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -3 | tee $OUT/bench_quant_bf16.txt | cut -c1-700
echo "== bench quantize f32 b=64"; timeout 600 python bench.py --quantize --dtype f32 --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-profile 2>&1 | tail -3 | tee $OUT/bench_quant_f32.txt | cut -c1-400
echo "== done"; date
