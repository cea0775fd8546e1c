#!/bin/bash
# PMC passes (own runs, kernel-trace only - never combined with sys/hip traces):
#  A: SQ pipeline counters on a few representative conv layers (what limits the MFMA kernels)
#  B/C: FETCH_SIZE / WRITE_SIZE over two ResNet-50 steps (HBM traffic per kernel for roofline.traffic)
export TMPDIR=/tmp
OUT=gpurun_out/${1:-pmc1}
mkdir -p $OUT
LAYERS="2,3,10,13,15,16,19,22"
run() { # name counters... -- cmd
  name=$1; shift; ctrs=$1; shift
  CONVNET_AMD_FLAGS=graph=0 timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/$name -o $name -- "$@" > $OUT/$name.log 2>&1
  echo "$name rc=$?"; ls $OUT/$name | head -5
}
run sq1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" python tools/bench_layers.py --iters 2 --variants 0 --only $LAYERS
run sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU" python tools/bench_layers.py --iters 2 --variants 0 --only $LAYERS
run fetch "FETCH_SIZE" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-issue-probe
run write "WRITE_SIZE" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-issue-probe
python tools/pmc_summary.py $OUT > $OUT/summary.txt; python tools/pmc_traffic.py $OUT $OUT/pmc_traffic.json | tee $OUT/traffic.txt
find $OUT -name "*.csv" -size +8M -delete
