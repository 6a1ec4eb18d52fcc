#!/bin/bash
# round 2: full GPU suite, default bench line + per-layer profile, the driver's exact command, rocprofv3 kernel stats of it,
# PMC passes (separate runs, kernel-trace only; the counter sets below are the ones that run on this image -- TCP_* / TA_* /
# TCC_EA0_* derived sums abort the process and were dropped)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 400 python bench.py --profile-out gpurun_out/bench_layers.json > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_cmd.json 2>/dev/null; echo "driver-cmd bench exit $?"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof -- python $R/bench.py --steps 20 --warmup 5 --prewarm-seconds 0.3 --no-cpu-baseline --no-also > $R/gpurun_out/rocprof_bench.json 2> $R/gpurun_out/rocprof.err
echo "rocprof exit $?"
find $R/gpurun_out/prof -name "*kernel_trace.csv" -delete
cd $R
bash tools/measure/gpu_pmc.sh \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
  "FETCH_SIZE TCC_HIT_sum SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
  "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" 2>&1 | tail -60
