#!/bin/bash
# Interleaved whole-step A/B on ONE box (how every block of profiles/r03_ab_second_session_whole_step.txt was made):
#   gpurun -- 'bash tools/ab_step.sh <tag> "<pytest -k filter or empty>" "<label A>" "<ENV=... for A>" "<label B>" "<ENV=... for B>" [rounds]'
# e.g. bash tools/ab_step.sh lazya "lazy_a" on X=1 off CONVNET_AMD_LAZY_A=0
# Runs the selected GPU tests first, one warm-up bench, then A / B alternately (bench.py --steps 60 --warmup 10);
# prints img/s, ms/step and the final loss per run and leaves them in gpurun_out/<tag>/sched.txt.
export TMPDIR=/tmp
OUT=gpurun_out/${1:?tag}; FILTER=$2; LA=${3:?label A}; EA=${4:?env A}; LB=${5:?label B}; EB=${6:?env B}; ROUNDS=${7:-3}
mkdir -p $OUT
if [ -n "$FILTER" ]; then
  timeout 900 python -m pytest tests -x -q -m gpu -k "$FILTER" 2>&1 | tail -3 | tee $OUT/pytest.txt
fi
run() { name=$1; shift; echo -n "$name: "; env "$@" timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-profile 2>&1 | grep '"metric"\|Error\|error' | head -2 | python -c "
import sys,json
t=sys.stdin.read()
try:
    d=json.loads(t); print(d['value'], d['ms_per_step'], d['config']['final_loss'])
except Exception: print('FAILED', t[:300])"; }
{
run warm X=1
for i in $(seq 1 $ROUNDS); do
run ${LA}_$i $EA
run ${LB}_$i $EB
done
} 2>&1 | tee $OUT/sched.txt
