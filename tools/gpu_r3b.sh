#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3b}
mkdir -p $OUT
echo "== warm parity + ilv tests"; timeout 900 python -m pytest tests/test_warm_parity.py tests/test_ops.py -m gpu -q --tb=short -p no:cacheprovider -s -k "warm or interleaved or 256x256" 2>&1 | grep -v "Warning\|warn\|detach\|total +=\|^$" | tail -40 | tee $OUT/pytest_warm.txt
echo "== A/B igemm_ilv"; timeout 600 python tools/bench_ab.py --knob igemm_ilv --values 0,1,2 --only 11,12,13,15,16,17,18,20,21,22 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_ilv.txt
echo "== marks=0 copyBuffer count"
CONVNET_AMD_MARKS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_nomarks -o r50 -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-profile > $OUT/rocprof_nomarks.log 2>&1
grep -h "copyBuffer\|fill" $(find $OUT/prof_nomarks -name "*kernel_stats.csv") | cut -c1-120
grep '"metric"' $OUT/rocprof_nomarks.log | cut -c1-200
find $OUT -name "*kernel_trace.csv" -size +20M -delete
echo "== done"; date
