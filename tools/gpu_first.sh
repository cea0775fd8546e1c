#!/bin/bash
# Opening set of a round: GPU suite, smoke, one bench line with the detail tables, rocprofv3 kernel stats.  tools/gpu_first.sh TAG
export TMPDIR=/tmp
OUT=gpurun_out/${1:-first}
mkdir -p $OUT
echo "== gpu tests"; timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -25 | tee $OUT/pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 --detail-out $OUT/bench_detail.json 2> $OUT/bench.err | tail -1 > $OUT/bench.json; cut -c1-300 $OUT/bench.json
echo "== rocprof"
CONVNET_AMD_FLAGS=graph=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r50 -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-profile > $OUT/rocprof.log 2>&1
STATS=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$STATS" ] && cp "$STATS" $OUT/kernel_stats.csv && head -12 $OUT/kernel_stats.csv | cut -c1-150
python tools/trace_by_grid.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) > $OUT/kernel_trace_by_grid.txt 2>&1
cp $(find $OUT/prof -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace.csv 2>/dev/null; gzip -f $OUT/kernel_trace.csv 2>/dev/null
rm -rf $OUT/prof
echo "== done"; date
