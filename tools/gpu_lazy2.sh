#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-lazy2}
mkdir -p $OUT
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"' > $OUT/bench_lazy1.json
CONVNET_AMD_LAZY_DY=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '"metric"' > $OUT/bench_lazy0.json
ls -la $OUT
