#!/bin/bash
# cache / texture-path counters on three GEMM-bound conv layers (variants 1 and 3)
export TMPDIR=/tmp
OUT=gpurun_out/${1:-pmc_cache}
mkdir -p $OUT
run() { name=$1; shift; ctrs=$1; shift
  timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/$name -o $name -- "$@" > $OUT/$name.log 2>&1
  echo "$name rc=$?"; }
CMD="python tools/bench_layers.py --iters 2 --variants 1,3 --fragdb 0 --only 15,16,22"
run tcc "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ" $CMD
run ta "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_BUFFER_READ_LDS_WAVEFRONTS TA_BUFFER_TOTAL_CYCLES GRBM_GUI_ACTIVE" $CMD
run tcp "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCP_TA_DATA_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES" $CMD
run lat "TCP_TCC_READ_REQ_LATENCY TCP_TCP_LATENCY TCP_TCC_READ_REQ TCP_TOTAL_READ" $CMD
python tools/pmc_summary.py $OUT | grep -E "== pass|^kernel|igemm" | cut -c1-400
