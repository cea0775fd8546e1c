#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r2b}
mkdir -p $OUT
echo "== new gpu tests"; timeout 1500 python -m pytest tests/test_graph_gpu.py tests/test_cli_and_dp.py tests/test_headline_parity.py tests/test_rccl_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -60 | tee $OUT/pytest_new.txt
echo "== trajectory + ops tests"; timeout 900 python -m pytest tests/test_trajectory.py tests/test_ops.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/pytest_traj.txt
B="--no-cpu-baseline --no-kernel-profile"
for i in 1 2; do
  echo "== graph on";  timeout 300 python bench.py --steps 30 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-230 | tee -a $OUT/ab_graph1.txt
  echo "== graph off"; CONVNET_AMD_GRAPH=0 timeout 300 python bench.py --steps 30 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-230 | tee -a $OUT/ab_graph0.txt
  echo "== 2 wgrad streams (graph on)"; CONVNET_AMD_WGRAD_STREAMS=2 timeout 300 python bench.py --steps 30 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-230 | tee -a $OUT/ab_ws2.txt
done
echo "== b=8 graph on"; timeout 300 python bench.py --batch 8 --steps 50 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-230 | tee $OUT/b8_graph1.txt
echo "== b=8 graph off"; CONVNET_AMD_GRAPH=0 timeout 300 python bench.py --batch 8 --steps 50 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-230 | tee $OUT/b8_graph0.txt
echo "== world-1 RCCL graph"; BENCH_FORCE_DIST=1 CONVNET_AMD_GRAPH_DP=1 timeout 300 python bench.py --steps 20 --warmup 5 $B 2>&1 | grep '"metric"' | cut -c80-230 | tee $OUT/dist1_graph.txt
echo "== done"; date
