#!/bin/bash
# round 2, final code: full GPU suite, the driver's exact command (+ per-layer profile), rocprofv3 kernel stats of it
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -3
timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --profile-out gpurun_out/bench_layers.json > gpurun_out/bench_driver_cmd.json 2>/dev/null; echo "driver-cmd bench exit $?"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof -- python $R/bench.py --steps 20 --warmup 5 --prewarm-seconds 0.3 --no-cpu-baseline --no-also > $R/gpurun_out/rocprof_bench.json 2> $R/gpurun_out/rocprof.err
echo "rocprof exit $?"
find $R/gpurun_out/prof -name "*kernel_trace.csv" -delete
cd $R; python -c "
import json
for f in ('gpurun_out/bench_driver_cmd.json',):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('all_mfma_conv'), d.get('also'))
"
