#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3f}
mkdir -p $OUT
echo "== A/B c64 tile"; timeout 600 python tools/bench_ab.py --knob igemm_c64_bm256 --values 0,1,2 --only 2,4 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_c64.txt
echo "== A/B ilv fragdb"; timeout 600 python tools/bench_ab.py --knob igemm_ilv_fragdb --values 0,1 --only 12,15,16,17,20 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_ilv_fragdb.txt
echo "== done"; date
