#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3c}
mkdir -p $OUT
echo "== warm debug"; timeout 900 python tools/warm_debug.py 2>&1 | grep -v "Warning\|warn\|detach\|total +=\|amdgpu.ids\|INFO\|^$" | tee $OUT/warm_debug.txt | tail -150
echo "== A/B igemm_dma_min_nkt"; timeout 600 python tools/bench_ab.py --knob igemm_dma_min_nkt --values 24,16,10,6 --rounds 3 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_min_nkt.txt
echo "== done"; date
