#!/bin/bash
# Config 5 (quantize=True): GPU tests of the quantised path, then bench.py --quantize A/B.  tools/gpu_c5.sh TAG ROUNDS specs...
export TMPDIR=/tmp
TAG=${1:-c5}; R=${2:-2}; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== quant tests"; timeout 1800 python -m pytest tests/test_quant.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/tests.txt
echo "== A/B (bench.py --quantize)"; AB_STEPS=10 AB_ARGS="--quantize" timeout 1800 python tools/gpu_ab.py $TAG $R "$@" 2>&1 | tee $OUT/ab.txt
echo "== detail"; timeout 600 python bench.py --quantize --steps 6 --warmup 3 --no-cpu-baseline --detail-out $OUT/bench_detail.json | tail -c 400
