#!/bin/bash
# Round-3 GPU call: parity tests, bench line (overlapped-step roofline), rocprofv3 kernel stats, per-layer table.
# usage: tools/gpu_r3.sh <tag> [tests|notests]
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3}
mkdir -p $OUT
{ date; python -c "import torch;print('torch',torch.__version__,'devices',torch.cuda.device_count(),torch.cuda.get_device_name(0))"; nproc; } > $OUT/env.txt 2>&1
if [ "${2:-tests}" = "tests" ]; then
  echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -60 | tee $OUT/pytest_gpu.txt
fi
echo "== bench"; timeout 900 python bench.py --steps ${BENCH_STEPS:-20} --warmup 5 2> $OUT/bench.err | grep '"metric"' > $OUT/bench.json; cut -c1-600 $OUT/bench.json; tail -3 $OUT/bench.err
echo "== layers"; timeout 600 python tools/bench_layers.py 2>&1 | grep -v amdgpu.ids | tail -30 | tee $OUT/layers.txt
echo "== rocprof"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r50 -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-profile > $OUT/rocprof.log 2>&1
STATS=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$STATS" ] && cp "$STATS" $OUT/kernel_stats.csv && head -25 $OUT/kernel_stats.csv | cut -c1-200
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
echo "== done"; date
