#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-sched6}
mkdir -p $OUT
run() { name=$1; shift; echo -n "$name: "; env "$@" timeout 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernel-profile 2>&1 | grep '"metric"\|Error\|error' | head -2 | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(d['value'], d['ms_per_step'], d['config']['final_loss'])
except Exception: print('FAILED', t[:300])"; }
{
run warm X=1
run base X=1
run epi8w CONVNET_AMD_OPTIONS=igemm_epi_8w=1
run base2 X=1
run epi8w_b CONVNET_AMD_OPTIONS=igemm_epi_8w=1
run epi8w_bm64 CONVNET_AMD_OPTIONS=igemm_epi_8w=1,igemm_epi_bm64=1
} 2>&1 | tee $OUT/sched.txt
echo "== parity with the knob"; CONVNET_AMD_OPTIONS=igemm_epi_8w=1 timeout 900 python -m pytest tests/test_ops.py tests/test_trajectory.py tests/test_headline_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "bwd or residual or bf16 or small" 2>&1 | tail -4
