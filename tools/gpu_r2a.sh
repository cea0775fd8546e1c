#!/bin/bash
# round-2 first GPU call: new parity / RCCL tests, bench line, NT A/B, rocprof summary
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r2a}
mkdir -p $OUT
{ date; python -c "import torch;print('torch',torch.__version__,'devices',torch.cuda.device_count(),torch.cuda.get_device_name(0))"; /opt/rocm/bin/rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock Freq|gfx" | head -8; nproc; free -g | head -2; } > $OUT/env.txt 2>&1
echo "== new gpu tests"; timeout 1500 python -m pytest tests/test_rccl_gpu.py tests/test_headline_parity.py tests/test_cli_and_dp.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -60 | tee $OUT/pytest_new.txt
echo "== all other gpu tests"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --deselect tests/test_rccl_gpu.py --deselect tests/test_headline_parity.py --deselect tests/test_cli_and_dp.py 2>&1 | tail -30 | tee $OUT/pytest_rest.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 5 2>&1 | grep '"metric"' | tee $OUT/bench.txt | cut -c1-400
for i in 1 2; do
  echo "== NT A/B bn_nt=0"; CONVNET_AMD_OPTIONS=bn_nt=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile 2>&1 | grep '"metric"' | cut -c1-200 | tee -a $OUT/ab_nt0.txt
  echo "== NT A/B bn_nt=1"; CONVNET_AMD_OPTIONS=bn_nt=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile 2>&1 | grep '"metric"' | cut -c1-200 | tee -a $OUT/ab_nt1.txt
done
echo "== world-1 direct RCCL"; BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile 2>&1 | grep '"metric"' | tee $OUT/bench_dist1.txt | cut -c1-500
echo "== rocprof"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r50 -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-profile > $OUT/rocprof.log 2>&1
STATS=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$STATS" ] && cp "$STATS" $OUT/kernel_stats.csv && head -25 "$STATS" | cut -c1-160
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
echo "== done"; date
