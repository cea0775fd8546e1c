#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r2l}
mkdir -p $OUT
echo "== quant tests"; timeout 900 python -m pytest tests/test_quant.py tests/test_gpu_probe.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -5 | tee $OUT/pytest_quant.txt
echo "== bench quantize bf16 b=256"; timeout 600 python bench.py --quantize --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_quant_bf16.txt | cut -c1-250
echo "== CLI: quantize config, 1 epoch of 3 steps"; timeout 600 python main.py --model resnet --model-config "{'depth': 18, 'quantize': True}" --dtype bfloat16 -b 32 --epochs 1 --steps-per-epoch 3 --val-steps 1 --print-freq 1 --results-dir /tmp/res_q --save q1 2>&1 | tail -6 | tee $OUT/cli_quant.txt
echo "== done"; date
