#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3d}
mkdir -p $OUT
echo "== band kernel tests"; timeout 900 python -m pytest tests/test_ops.py -m gpu -q --tb=short -p no:cacheprovider -k "wgrad or interleaved" 2>&1 | grep -v "Warning\|warn\|^$" | tail -15 | tee $OUT/pytest_wgrad.txt
echo "== A/B wgrad_3x3"; timeout 600 python tools/bench_ab.py --knob wgrad_3x3 --values 0,1 --dirs wgrad --only 2,10,16,22 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_wgrad3x3.txt
echo "== A/B wgrad_3x3_wgs"; timeout 600 python tools/bench_ab.py --knob wgrad_3x3_wgs --values 256,512,768,1024 --dirs wgrad --only 2,10,16,22 --rounds 3 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_wgrad3x3_wgs.txt
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> $OUT/bench.err | grep '"metric"' > $OUT/bench.json; cut -c1-330 $OUT/bench.json
echo "== bench wgrad_3x3=0"; CONVNET_AMD_OPTIONS="wgrad_3x3=0" timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile 2>> $OUT/bench.err | grep '"metric"' | cut -c1-330 | tee $OUT/bench_nowg3.json
echo "== done"; date
