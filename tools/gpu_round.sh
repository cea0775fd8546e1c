#!/bin/bash
# One gpurun call: environment facts, lane-map probes, GPU parity tests, bench line, rocprof summary.
# Everything lands under gpurun_out/ (merged back by gpurun).
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r1}
mkdir -p $OUT
{
  echo "== env"; date
  python -c "import torch;print('torch',torch.__version__,'devices',torch.cuda.device_count(),torch.cuda.get_device_name(0))"
  /opt/rocm/bin/rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock Freq|gfx" | head -12
  python -c "import sys;sys.path.insert(0,'.');from oracle.convnet_oracle import usable_cpus;print('usable cpus',usable_cpus())"; cat /sys/fs/cgroup/cpu.max
  nproc; grep -m1 "model name" /proc/cpuinfo; free -g | head -2
} > $OUT/env.txt 2>&1
echo "== probes" ; timeout 600 python -m pytest tests/test_gpu_probe.py -m gpu -q --tb=short 2>&1 | tail -30 | tee $OUT/probe.txt
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -120 | tee $OUT/pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -15 | tee $OUT/smoke.txt
echo "== layers"; timeout 600 python tools/bench_layers.py 2>&1 | tail -30 | tee $OUT/layers.txt
echo "== bench"; timeout 900 python bench.py --steps ${BENCH_STEPS:-10} --warmup 3 2>&1 | tail -5 | tee $OUT/bench.txt
echo "== host overhead (B=8: GPU work is small, step time ~ host launch cost)"; timeout 300 python bench.py --batch 8 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile 2>&1 | grep '"metric"' | tee $OUT/bench_b8.txt | cut -c1-330
echo "== PCIe-inclusive rate (pinned host batches, H2D inside the timed loop; never the contract value)"; timeout 300 python bench.py --host-inputs --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-profile 2>&1 | grep '"metric"' | tee $OUT/bench_host_inputs.txt | cut -c1-330
echo "== world-1 RCCL path"; BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-profile 2>&1 | grep '"metric"' | tee $OUT/bench_dist1.txt | cut -c1-330
echo "== ResNet-18 fp32 b=256 (BASELINE config 1, parity-test case)"; timeout 600 python bench.py --depth 18 --dtype f32 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep '"metric"' | tee $OUT/bench_r18_f32.txt | cut -c1-400
echo "== epilogue fusions per layer"; timeout 600 python tools/bench_fusion.py 2>&1 | grep -v amdgpu.ids | tail -26 | tee $OUT/fusion.txt
echo "== vendor GEMM library on the same GEMM shapes (ceiling reference, not a product path)"; timeout 300 python tools/gemm_ceiling.py 2>&1 | grep -v amdgpu.ids | tail -10 | tee $OUT/gemm_ceiling.txt
echo "== rocprof"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o r50 -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-profile > $OUT/rocprof.log 2>&1
ls -R $OUT/prof | head -30
STATS=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$STATS" ] && head -40 "$STATS" | tee $OUT/kernel_stats_head.csv
# keep only the small summaries (the raw trace can be large)
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
echo "== done"; date
