#!/bin/bash
# round 2: default bench line + per-layer profile, rocprofv3 kernel stats of the driver's command, PMC passes (separate runs,
# kernel-trace only): instruction mix / waits, L1<->L2 request latency, L2 hit + fabric requests, TA/TD stalls, HBM bytes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 600 python bench.py --profile-out gpurun_out/bench_layers.json > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof -- python $R/bench.py --steps 20 --warmup 5 --prewarm-seconds 0.3 --no-cpu-baseline --no-also > $R/gpurun_out/rocprof_bench.json 2> $R/gpurun_out/rocprof.err
echo "rocprof exit $?"
find $R/gpurun_out/prof -name "*kernel_trace.csv" -delete
cd $R
bash tools/measure/gpu_pmc.sh \
  "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
  "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE" \
  "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE" \
  "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum TD_TD_BUSY_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
  "FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
  "WRITE_SIZE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" 2>&1 | tail -150
