#!/bin/bash
# rocprofv3 kernel traces of the eager step and of the launch plan (tools/plan_probe.py), compared per kernel name.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06}
mkdir -p $O
cd /tmp
for m in eager plan; do
  rm -rf /tmp/prof_$m
  timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$m -o t -- python $R/tools/plan_probe.py --modes $m --rounds 1 --steps 14 > $O/trace_$m.log 2>&1
  tail -3 $O/trace_$m.log
done
cd $R
python tools/trace_cmp_modes.py $(find /tmp/prof_eager -name "*kernel_trace.csv" | head -1) $(find /tmp/prof_plan -name "*kernel_trace.csv" | head -1) 0.4 | tee $O/trace_cmp_eager_vs_plan.txt
python tools/trace_step_timeline.py $(find /tmp/prof_eager -name "*kernel_trace.csv" | head -1) $(find /tmp/prof_plan -name "*kernel_trace.csv" | head -1) | tee $O/trace_timeline_eager_vs_plan.txt
