#!/bin/bash
# full parity suite + default bench line + rocprofv3 kernel stats + PMC passes (separate runs)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q --tb=short 2>&1 | grep -v amdgpu.ids | tail -8
timeout 600 python bench.py --profile-out gpurun_out/bench_layers.json > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"
cat gpurun_out/bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof -- python $R/bench.py --steps 20 --warmup 5 --prewarm-seconds 0.3 --no-cpu-baseline --no-also > $R/gpurun_out/rocprof_bench.json 2> $R/gpurun_out/rocprof.err
echo "rocprof exit $?"
find $R/gpurun_out/prof -name "*kernel_trace.csv" -delete
cd $R
[ -n "$NO_PMC" ] || bash tools/measure/gpu_pmc.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE" "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" 2>&1 | tail -40
