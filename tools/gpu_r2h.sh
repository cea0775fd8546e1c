#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r2h}
mkdir -p $OUT
echo "== probe + quant + abi tests"; timeout 900 python -m pytest tests/test_gpu_probe.py tests/test_quant.py tests/test_abi.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -30 | tee $OUT/pytest_quant.txt
echo "== bench quantize bf16 b=256, float kernels"; timeout 600 python bench.py --quantize --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_quant_bf16.txt | cut -c1-250
echo "== bench quantize bf16 b=256, int8 MFMA forward"; CONVNET_AMD_QUANT_INT8=1 timeout 600 python bench.py --quantize --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_quant_bf16_int8.txt | cut -c1-250
echo "== done"; date
