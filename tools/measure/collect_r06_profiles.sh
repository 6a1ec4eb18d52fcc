#!/bin/bash
# copy the round-6 final script's artefacts (gpurun_out/r06final, gpurun_out/r06pmc) into profiles/ under their committed names
cd "$(dirname "$0")/../.."; F=gpurun_out/r06final; P=gpurun_out/r06pmc
cp $F/kernel_sources_sha256.txt profiles/r06_kernel_sources_sha256.txt
cp $F/source_commit.txt profiles/r06_source_commit.txt
cp $F/pytest_gpu.txt profiles/r06_pytest_gpu.txt; cp $F/smoke.txt profiles/r06_smoke.txt
cp $F/pmc_calibration.json profiles/pmc_calibration.json
cp $F/pmc_by_kernel.json profiles/r06_pmc_by_kernel.json
cp $F/pmc_traffic_sharp_b8_f16.json profiles/pmc_traffic_sharp_b8_f16.json
cp $F/rocprofv3_kernel_stats_sharp_b8_f16.json profiles/rocprofv3_kernel_stats_sharp_b8_f16.json
cp $F/rocprofv3_kernel_stats_sharp_b8_f16.csv profiles/r06_rocprofv3_kernel_stats.csv
cp $F/rocprofv3_kernel_stats_sharp_b8_f32.csv profiles/r06_rocprofv3_kernel_stats_f32.csv
cp $F/bench_driver_cmd.json profiles/r06_bench_driver_cmd.json
cp $F/bench_b8_f32.json profiles/r06_bench_b8_f32.json; cp $F/bench_b64.json profiles/r06_bench_b64.json; cp $F/bench_b1.json profiles/r06_bench_b1.json
cp $F/layers_b8.json profiles/r06_layers_b8.json; cp $F/layers_b64.json profiles/r06_layers_b64.json; cp $F/layers_b1.json profiles/r06_layers_b1.json
cp $F/b64_kernel_table.json profiles/r06_b64_kernel_table.json; cp $F/b1_kernel_table.json profiles/r06_b1_kernel_table.json
cp $F/b8_f32_kernel_table.json profiles/r06_b8_f32_kernel_table.json
cp $F/b8_f16x3_kernel_table.json profiles/r06_b8_f16x3_kernel_table.json; cp $F/bench_b8_f16x3.json profiles/r06_bench_b8_f16x3.json; cp $F/layers_b8_f16x3.json profiles/r06_layers_b8_f16x3.json
cp $F/argmax_agreement_f16x3.json profiles/r06_argmax_agreement_f16x3.json
for t in f16x3 f32 f16; do cp $F/tools_on_mi355x_$t.json profiles/r06_tools_on_mi355x_$t.json 2>/dev/null; done
cp $F/seq_phase_clocks.txt profiles/r06_seq_phase_clocks.txt
cp $F/pipelined_step_timeline.txt profiles/r06_pipelined_step_timeline.txt
cp $F/dry_run.json profiles/r06_dry_run.json
cp $F/argmax_agreement.json profiles/r06_argmax_agreement.json
ls -la profiles/r06_* | wc -l
