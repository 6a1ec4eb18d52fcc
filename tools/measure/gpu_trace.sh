#!/bin/bash
# rocprofv3 kernel trace of the bench in graph-replay mode -> per-kernel durations and gaps
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
python bench.py --steps 50 --warmup 10 --no-cpu-baseline ${ALSO:---no-also} --tune "${TUNE:-stages=0}" --profile-out gpurun_out/bench_layers.json > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json | head -c 1500; echo
cd /tmp && export TMPDIR=/tmp; rm -rf $R/gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --tune "${TUNE:-stages=0}" > $R/gpurun_out/rocprof_bench.json 2> $R/gpurun_out/rocprof.err
cd $R
python tools/measure/trace_stats.py
find gpurun_out/prof -name "*kernel_trace.csv" -size +8M -delete
