#!/bin/bash
# PMC passes (separate runs, kernel-trace only -- never with sys/hip traces) for the bench step.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/pmc
rocprofv3 -L > gpurun_out/pmc/counters_list.txt 2>&1
# --serial: counter collection serialises the dispatches of ALL queues, which the pipelined step's gate kernels cannot survive (a gate waits
# for a kernel of the other queue that the profiler will not start before the gate ends: 0.2 s time-outs, the persistent launches behind the
# raised flag return at once and pollute every per-kernel average -- seen in round 5's first final run: 159 MB "per launch" instead of 647)
CMD="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-also --no-long --serial --prewarm-seconds 0.3 ${WL:+--workload $WL}"
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "$@"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc/pass$i
  timeout 150 rocprofv3 --pmc $SET --kernel-trace -f csv -d $R/gpurun_out/pmc/pass$i -- $CMD > $R/gpurun_out/pmc/pass$i.out 2> $R/gpurun_out/pmc/pass$i.err
  echo "pass $i [$SET] exit $?"
done
cd $R
python tools/measure/pmc_stats.py gpurun_out/pmc
# drop raw traces that are too large to bring back
find gpurun_out/pmc -name "*.csv" -size +12M -delete
