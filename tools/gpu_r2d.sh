#!/bin/bash
# round-2 baseline call: full GPU suite, bench line, rocprofv3 stats, per-layer table
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r2d}
mkdir -p $OUT
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -30 | tee $OUT/pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.txt
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -3 | tee $OUT/bench.txt | cut -c1-600
echo "== layers"; timeout 600 python tools/bench_layers.py 2>&1 | tail -30 | tee $OUT/layers.txt
echo "== rocprof"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r50 -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-profile > $OUT/rocprof.log 2>&1
STATS=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$STATS" ] && head -40 "$STATS" | tee $OUT/kernel_stats_head.csv
TR=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
[ -n "$TR" ] && python tools/trace_by_grid.py "$TR" > $OUT/trace_by_grid.txt 2>&1
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
echo "== done"; date
