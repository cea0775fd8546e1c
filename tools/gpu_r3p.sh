#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3p}
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_quant.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/pytest.txt
for cfg in "auto 1" "0 1" "auto 0" "1 1"; do set -- $cfg; echo -n "GRAPH=$1 FUSE_RBN=$2: "; CONVNET_AMD_GRAPH=$1 CONVNET_AMD_QUANT_FUSE_RBN=$2 timeout 600 python bench.py --quantize --steps 10 --warmup 8 --no-cpu-baseline --no-kernel-profile 2>$OUT/err_$1_$2.txt | grep '"metric"' | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(d['value'], d['ms_per_step'], d['config']['final_loss'])
except Exception: print('FAILED')"; tail -2 $OUT/err_$1_$2.txt | cut -c1-200; done 2>&1 | tee $OUT/sched.txt
