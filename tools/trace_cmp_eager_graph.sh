# rocprofv3 kernel trace of the same short bench run, eager (graph=0) and replayed (graph=1): stats, gaps, per-grid times and
# the last three steps as (queue, kernel, start, end) rows under gpurun_out/r4tr/ (which hardware queues each mode uses).
export TMPDIR=/tmp
OUT=gpurun_out/r4tr
mkdir -p $OUT
for mode in eager graph; do
  F="graph=0"; [ $mode = graph ] && F="graph=1"
  CONVNET_AMD_FLAGS=$F timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$mode -o r50 -- python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-kernel-profile > $OUT/$mode.log 2>&1
  tail -1 $OUT/$mode.log | cut -c1-200
  TR=$(find $OUT/$mode -name "*kernel_trace.csv" | head -1)
  ST=$(find $OUT/$mode -name "*kernel_stats.csv" | head -1)
  cp $ST $OUT/${mode}_stats.csv
  python tools/trace_gaps.py $TR > $OUT/${mode}_gaps.txt 2>&1
  python tools/trace_by_grid.py $TR > $OUT/${mode}_by_grid.txt 2>&1
  # keep the last two steps of the trace only (size)
  python - "$TR" "$OUT/${mode}_last_steps.csv" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
marks=[i for i,r in enumerate(rows) if 'nchw_to_pairs' in r['Kernel_Name']]
a=marks[-3] if len(marks)>=3 else 0
w=csv.writer(open(sys.argv[2],'w'))
w.writerow(['q','name','grid','start','end'])
for r in rows[a:]:
    w.writerow([r['Queue_Id'],r['Kernel_Name'][:60],r.get('Grid_Size_X',r.get('Grid_Size','')),r['Start_Timestamp'],r['End_Timestamp']])
PY
  find $OUT/$mode -name "*.csv" -size +5M -delete
done
head -5 $OUT/eager_gaps.txt; head -5 $OUT/graph_gaps.txt
