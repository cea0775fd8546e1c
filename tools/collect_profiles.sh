#!/bin/bash
# Copy the closing-set artefacts of a gpurun_out/<tag>/ directory (tools/gpu_final.sh) into profiles/ under <rNN>_ names.
set -e
S=gpurun_out/${1:?tag}; P=profiles; R=${2:?round tag, e.g. r04}
cp $S/bench.json $P/${R}_bench_line.json
cp $S/bench_detail.json $P/${R}_bench_detail.json
cp $S/bench_b8.json $P/${R}_bench_line_b8.json
cp $S/bench_config5.json $P/${R}_bench_line_config5_quantize_bf16.json
cp $S/bench_dist1.json $P/${R}_bench_line_world1_rccl.json
cp $S/bench_host_inputs.json $P/${R}_bench_line_host_inputs_pcie.json
cp $S/bench_r101.json $P/${R}_bench_line_resnet101_bf16.json
cp $S/bench_r18_f32.json $P/${R}_bench_line_resnet18_fp32.json
cp $S/env.txt $P/${R}_env.txt
cp $S/kernel_stats.csv $P/${R}_rocprofv3_kernel_stats_resnet50_bf16_b256.csv
cp $S/kernel_trace_by_grid.txt $P/${R}_kernel_trace_by_grid.txt
cp $S/layers.txt $P/${R}_conv_layers_b256_bf16.txt
cp $S/pmc_traffic.json $P/${R}_pmc_traffic.json
for f in plan_probe_b256.json plan_probe_b8.json; do [ -f $S/$f ] && cp $S/$f $P/${R}_$f; done
[ -f $S/c5_kernel_stats.csv ] && cp $S/c5_kernel_stats.csv $P/${R}_rocprofv3_kernel_stats_config5_quantize_bf16.csv
if [ -f $S/mfma_step.txt ]; then cp $S/mfma_step.txt $P/${R}_pmc_mfma_utilisation_per_kernel.txt; fi
{ cat $S/pytest_gpu.txt; echo; echo "== smoke"; cat $S/smoke.txt; } > $P/${R}_pytest_gpu_tail.txt
ls -la $P | grep ${R}_ | wc -l
