#!/bin/bash
# round-2 closing call: full GPU suite, smoke, bench lines, rocprofv3 stats + per-grid trace, per-layer table, PMC passes
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r2final}
mkdir -p $OUT
{
  date; python -c "import torch;print('torch',torch.__version__,'devices',torch.cuda.device_count(),torch.cuda.get_device_name(0))"
  /opt/rocm/bin/rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock Freq|gfx" | head -8
  python -c "import sys;sys.path.insert(0,'.');from oracle.convnet_oracle import usable_cpus;print('usable cpus',usable_cpus())"
} > $OUT/env.txt 2>&1
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | tee $OUT/bench.txt | cut -c1-260
B="--no-cpu-baseline --no-kernel-profile"
echo "== bench again (no profile pass)"; timeout 300 python bench.py --steps 30 --warmup 5 $B 2>&1 | grep '"metric"' | tee $OUT/bench_plain.txt | cut -c80-200
echo "== b=8 (host overhead / graph)"; timeout 300 python bench.py --batch 8 --steps 50 --warmup 5 $B 2>&1 | grep '"metric"' | tee $OUT/bench_b8.txt | cut -c80-200
echo "== host inputs (PCIe-inclusive)"; timeout 300 python bench.py --host-inputs --steps 20 --warmup 5 $B 2>&1 | grep '"metric"' | tee $OUT/bench_host_inputs.txt | cut -c80-200
echo "== world-1 RCCL"; BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 20 --warmup 5 $B 2>&1 | grep '"metric"' | tee $OUT/bench_dist1.txt | cut -c80-200
echo "== ResNet-18 fp32"; timeout 600 python bench.py --depth 18 --dtype f32 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep '"metric"' | tee $OUT/bench_r18_f32.txt | cut -c80-200
echo "== ResNet-101 bf16"; timeout 600 python bench.py --depth 101 --steps 5 --warmup 2 $B 2>&1 | grep '"metric"' | tee $OUT/bench_r101.txt | cut -c80-200
echo "== config 5 (quantize=True)"; timeout 600 python bench.py --quantize --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '"metric"' | tee $OUT/bench_quantize.txt | cut -c80-220
echo "== layers"; timeout 600 python tools/bench_layers.py --variants 0,1,3 2>&1 | tail -26 | tee $OUT/layers.txt | tail -2
echo "== rocprof"   # eager launches forced: under the tracer the trainer's auto mode would replay the step as a HIP graph
CONVNET_AMD_GRAPH=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r50 -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-profile > $OUT/rocprof.log 2>&1
STATS=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$STATS" ] && head -12 "$STATS" | cut -c1-160
TR=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
[ -n "$TR" ] && python tools/trace_by_grid.py "$TR" > $OUT/trace_by_grid.txt 2>&1
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
echo "== pmc"; bash tools/pmc_round.sh ${1:-r2final}/pmc 2>&1 | tail -12
echo "== done"; date
