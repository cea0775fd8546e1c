#!/bin/bash
# round 3, second session: dual-BN junction + lazy z - GPU parity tests, then interleaved whole-step A/B on one box
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3b}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops.py tests/test_trajectory.py tests/test_graph_gpu.py tests/test_step_local_consistency.py \
  -x -q -m gpu -k "dual or lazy_z or lazy_dy or graph or local or batchnorm_train" 2>&1 | tail -15 | tee $OUT/pytest.txt
run() { name=$1; shift; echo -n "$name: "; env "$@" timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-profile 2>&1 | grep '"metric"\|Error\|error' | head -2 | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(d['value'], d['ms_per_step'], d['config']['final_loss'])
except Exception: print('FAILED', t[:300])"; }
{
run warm X=1
for i in 1 2 3; do
run base_$i CONVNET_AMD_DUAL_BN=0 CONVNET_AMD_LAZY_Z=0
run dual_$i CONVNET_AMD_DUAL_BN=1 CONVNET_AMD_LAZY_Z=0
run lazyz_$i CONVNET_AMD_DUAL_BN=0 CONVNET_AMD_LAZY_Z=1
run both_$i CONVNET_AMD_DUAL_BN=1 CONVNET_AMD_LAZY_Z=1
done
} 2>&1 | tee $OUT/sched.txt
