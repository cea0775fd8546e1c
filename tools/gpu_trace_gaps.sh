#!/bin/bash
# idle gaps between consecutive kernels per queue in a launch-plan trace (tools/trace_gaps.py)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06}
mkdir -p $O
cd /tmp; rm -rf /tmp/prof_gap
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_gap -o t -- python $R/bench.py --steps 6 --warmup 5 --no-cpu-baseline --no-kernel-profile > $O/gaps.log 2>&1
cd $R
python tools/trace_gaps.py $(find /tmp/prof_gap -name "*kernel_trace.csv" | head -1) | tee $O/trace_gaps_plan.txt
python tools/trace_boundary.py $(find /tmp/prof_gap -name "*kernel_trace.csv" | head -1) 450 | tee $O/trace_boundary_plan.txt
