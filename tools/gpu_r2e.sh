#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r2e}
mkdir -p $OUT
echo "== layers big tiles"; timeout 900 python tools/bench_layers.py --variants 0,7,8,9,10 --iters 10 2>&1 | tail -28 | tee $OUT/layers_bigtile.txt
echo "== done"; date
