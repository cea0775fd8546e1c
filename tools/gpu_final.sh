#!/bin/bash
# Closing set of a round (tools/gpu_final.sh TAG; then tools/collect_profiles.sh TAG rNN): environment, GPU tests, smoke, bench line (+ live PMC traffic), rocprofv3 kernel stats, per-layer
# tables, the secondary bench lines.  Everything lands under gpurun_out/<tag>/.
export TMPDIR=/tmp
OUT=gpurun_out/${1:-final}
mkdir -p $OUT
{ date; python -c "import torch;print('torch',torch.__version__,'devices',torch.cuda.device_count(),torch.cuda.get_device_name(0))"
  /opt/rocm/bin/rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock Freq|gfx" | head -8
  python -c "import sys;sys.path.insert(0,'.');from oracle.convnet_oracle import usable_cpus;print('usable cpus',usable_cpus())"; nproc; } > $OUT/env.txt 2>&1
echo "== gpu tests"; timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -40 | tee $OUT/pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.txt
echo "== bench (+pmc)"; timeout 1800 python bench.py --steps 20 --warmup 5 --pmc --pmc-out $OUT/pmc_traffic.json --detail-out $OUT/bench_detail.json 2> $OUT/bench.err | tail -1 > $OUT/bench.json; cut -c1-400 $OUT/bench.json; wc -c $OUT/bench.json; tail -3 $OUT/bench.err
echo "== layers"; timeout 600 python tools/bench_layers.py --variants 0 2>&1 | grep -v amdgpu.ids | tail -30 | tee $OUT/layers.txt
echo "== b8"; timeout 300 python bench.py --batch 8 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile 2>/dev/null | grep '"metric"' | tee $OUT/bench_b8.json | cut -c1-250
echo "== host inputs"; timeout 300 python bench.py --host-inputs --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile 2>/dev/null | grep '"metric"' | tee $OUT/bench_host_inputs.json | cut -c1-250
echo "== world-1 RCCL"; BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile 2>/dev/null | grep '"metric"' | tee $OUT/bench_dist1.json | cut -c1-300
echo "== R18 fp32"; timeout 600 python bench.py --depth 18 --dtype f32 --steps 6 --warmup 4 --no-cpu-baseline 2>/dev/null | grep '"metric"' | tee $OUT/bench_r18_f32.json | cut -c1-250
echo "== R101 bf16"; timeout 600 python bench.py --depth 101 --steps 20 --warmup 10 --no-cpu-baseline --no-kernel-profile 2>/dev/null | grep '"metric"' | tee $OUT/bench_r101.json | cut -c1-250
echo "== config 5"; timeout 600 python bench.py --quantize --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '"metric"' | tee $OUT/bench_config5.json | cut -c1-250
echo "== rocprof"
# (default flags: after four eager warm-up steps the traced steps are launch-plan replays - the shipped step)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r50 -- python bench.py --steps 6 --warmup 5 --no-cpu-baseline --no-kernel-profile > $OUT/rocprof.log 2>&1
STATS=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$STATS" ] && cp "$STATS" $OUT/kernel_stats.csv && head -30 $OUT/kernel_stats.csv | cut -c1-170
python tools/trace_by_grid.py $(find $OUT/prof -name "*kernel_trace.csv" | head -1) > $OUT/kernel_trace_by_grid.txt 2>&1
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
echo "== MFMA utilisation from SQ counters"; timeout 900 python tools/pmc_mfma_step.py $OUT 2>&1 | tail -32
rm -rf $OUT/sqstep
echo "== launch plan vs eager vs HIP graph"; timeout 600 python tools/plan_probe.py --out $OUT/plan_probe_b256.json 2>&1 | grep -E "step|plan\[" ; timeout 300 python tools/plan_probe.py --batch 8 --steps 100 --out $OUT/plan_probe_b8.json 2>&1 | grep step
echo "== config 5 kernel stats"; tools/gpu_c5_stats.sh $(basename $OUT) "" "" | tail -30
echo "== done"; date
