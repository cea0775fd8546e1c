#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r2c}
mkdir -p $OUT
echo "== tests"; timeout 1500 python -m pytest tests/test_graph_gpu.py tests/test_headline_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "graph or batchnorm" 2>&1 | tail -8 | tee $OUT/pytest_new.txt
B="--no-cpu-baseline --no-kernel-profile"
for i in 1 2; do
  echo "== auto";  timeout 300 python bench.py --steps 30 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-200 | tee -a $OUT/ab_auto.txt
  echo "== 2 wgrad streams"; CONVNET_AMD_WGRAD_STREAMS=2 timeout 300 python bench.py --steps 30 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-200 | tee -a $OUT/ab_ws2.txt
  echo "== 3 wgrad streams"; CONVNET_AMD_WGRAD_STREAMS=3 timeout 300 python bench.py --steps 30 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-200 | tee -a $OUT/ab_ws3.txt
done
echo "== b=8 auto"; timeout 300 python bench.py --batch 8 --steps 50 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-200 | tee $OUT/b8_auto.txt
echo "== b=32 auto"; timeout 300 python bench.py --batch 32 --steps 50 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-200 | tee $OUT/b32_auto.txt
echo "== b=32 off"; CONVNET_AMD_GRAPH=0 timeout 300 python bench.py --batch 32 --steps 50 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-200 | tee $OUT/b32_off.txt
echo "== layers"; timeout 600 python tools/bench_layers.py --variants 0 2>&1 | tail -28 | tee $OUT/layers.txt
echo "== done"; date
