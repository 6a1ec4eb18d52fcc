#!/bin/bash
# copy the round-5 final script's artefacts (gpurun_out/r05final, gpurun_out/r05pmc) into profiles/ under their committed names
cd "$(dirname "$0")/../.."; F=gpurun_out/r05final; P=gpurun_out/r05pmc
cp $F/kernel_sources_sha256.txt profiles/r05_kernel_sources_sha256.txt
cp $F/source_commit.txt profiles/r05_source_commit.txt
cp $F/pytest_gpu.txt profiles/r05_pytest_gpu.txt; cp $F/smoke.txt profiles/r05_smoke.txt
cp $F/pmc_calibration.json profiles/pmc_calibration.json
cp $F/pmc_by_kernel.json profiles/r05_pmc_by_kernel.json
cp $F/pmc_traffic_sharp_b8_f16.json profiles/pmc_traffic_sharp_b8_f16.json
cp $F/rocprofv3_kernel_stats_sharp_b8_f16.json profiles/rocprofv3_kernel_stats_sharp_b8_f16.json
cp $F/rocprofv3_kernel_stats_sharp_b8_f16.csv profiles/r05_rocprofv3_kernel_stats.csv
cp $F/rocprofv3_kernel_stats_sharp_b8_f32.csv profiles/r05_rocprofv3_kernel_stats_f32.csv
cp $F/bench_driver_cmd.json profiles/r05_bench_driver_cmd.json
cp $F/bench_b8_f32.json profiles/r05_bench_b8_f32.json; cp $F/bench_b64.json profiles/r05_bench_b64.json; cp $F/bench_b1.json profiles/r05_bench_b1.json
cp $F/layers_b8.json profiles/r05_layers_b8.json; cp $F/layers_b64.json profiles/r05_layers_b64.json; cp $F/layers_b1.json profiles/r05_layers_b1.json
cp $F/b64_kernel_table.json profiles/r05_b64_kernel_table.json; cp $F/b1_kernel_table.json profiles/r05_b1_kernel_table.json
cp $F/b8_f32_kernel_table.json profiles/r05_b8_f32_kernel_table.json
cp $F/seq_phase_clocks.txt profiles/r05_seq_phase_clocks.txt
cp $F/pipelined_step_timeline.txt profiles/r05_pipelined_step_timeline.txt
cp $F/dry_run.json profiles/r05_dry_run.json
cp $F/argmax_agreement.json profiles/r05_argmax_agreement.json
ls -la profiles/r05_* | wc -l
