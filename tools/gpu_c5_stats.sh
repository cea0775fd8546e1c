#!/bin/bash
# rocprofv3 kernel stats of the config-5 step (bench.py --quantize), eager launches; optional CONVNET_AMD_FLAGS in $2
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06}
mkdir -p $O
cd /tmp
rm -rf /tmp/prof_c5
CONVNET_AMD_FLAGS="graph=0${2:+,$2}" timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -o c5 -- python $R/bench.py --quantize --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-profile > $O/c5_rocprof.log 2>&1
tail -1 $O/c5_rocprof.log | cut -c1-400
S=$(find /tmp/prof_c5 -name "*kernel_stats.csv" | head -1)
cp $S $O/c5_kernel_stats${3}.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$S")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms over the run: %.1f' % (tot/1e6))
for r in rows[:32]:
    print('%-100s %6s %9.2f ms %8.1f us %5.1f%%' % (r['Name'][:100], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
