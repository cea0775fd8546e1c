#!/bin/bash
# whole-step A/B of scheduling knobs (each line: one bench.py run, eager, no profile pass)
export TMPDIR=/tmp
OUT=gpurun_out/${1:-sched}
mkdir -p $OUT
run() { name=$1; shift; echo -n "$name: "; env "$@" timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-profile 2>/dev/null | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
{
run base X=1
run base2 X=1
run side_prio_low CONVNET_AMD_WGRAD_STREAM_PRIO=0
run side_prio_high CONVNET_AMD_WGRAD_STREAM_PRIO=-1
run main_prio_norm CONVNET_AMD_MAIN_STREAM_PRIO=0
run two_side CONVNET_AMD_WGRAD_STREAMS=2
run no_side CONVNET_AMD_WGRAD_STREAM=0
run wg3_off CONVNET_AMD_OPTIONS=wgrad_3x3=0
run wg3_256 CONVNET_AMD_OPTIONS=wgrad_3x3_wgs=256
run ilv0 CONVNET_AMD_OPTIONS=igemm_ilv=0
run base3 X=1
} 2>&1 | tee $OUT/sched.txt
