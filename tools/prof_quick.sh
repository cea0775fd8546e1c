#!/bin/bash
# quick rocprofv3 kernel-stats pass over a short bench run
export TMPDIR=/tmp
OUT=gpurun_out/${1:-pq}
mkdir -p $OUT
timeout 600 env CONVNET_AMD_FLAGS=wgrad_stream=${WGRAD_STREAM:-1} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r50 -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-profile > $OUT/rocprof.log 2>&1
STATS=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$STATS" ] && cp "$STATS" $OUT/kernel_stats.csv && head -30 "$STATS" | cut -c1-150
TR=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); [ -n "$TR" ] && python tools/trace_by_grid.py $TR > $OUT/by_grid.txt; find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
