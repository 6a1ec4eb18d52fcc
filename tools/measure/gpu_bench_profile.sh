#!/bin/bash
# smoke + bench (with per-layer launch profile) + rocprofv3 kernel trace of the same command.
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd $R
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
python bench.py --steps ${STEPS:-50} --warmup 10 --profile-out gpurun_out/bench_layers.json > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench exit $?" >> gpurun_out/bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also > $R/gpurun_out/rocprof_bench.json 2> $R/gpurun_out/rocprof.err
echo "rocprof exit $?" >> $R/gpurun_out/rocprof.err
# keep only the small summaries (the raw kernel trace can be large)
find $R/gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
cd $R; tail -3 gpurun_out/smoke.log; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
