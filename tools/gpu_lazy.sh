#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-lazy}
mkdir -p $OUT
echo "== lazy tests"; timeout 900 python -m pytest tests/test_ops.py tests/test_trajectory.py -m gpu -q --tb=short -p no:cacheprovider -k "lazy_dy" 2>&1 | grep -v "Warning\|warn\|^$" | tail -8 | tee $OUT/pytest_lazy.txt
run() { name=$1; shift; echo -n "$name: "; env "$@" timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-profile 2>&1 | grep '"metric"\|Error\|error' | head -2 | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(d['value'], d['ms_per_step'])
except Exception: print('FAILED', t[:300])"; }
{
run lazy0 CONVNET_AMD_LAZY_DY=0
run lazy150 CONVNET_AMD_LAZY_DY_MIN_MB=150
run lazy300 CONVNET_AMD_LAZY_DY_MIN_MB=300
run lazy90 CONVNET_AMD_LAZY_DY_MIN_MB=90
run lazy45 CONVNET_AMD_LAZY_DY_MIN_MB=45
run lazy0mb CONVNET_AMD_LAZY_DY_MIN_MB=0
run lazy0_b CONVNET_AMD_LAZY_DY=0
run lazy150_b CONVNET_AMD_LAZY_DY_MIN_MB=150
} 2>&1 | tee $OUT/sched.txt
