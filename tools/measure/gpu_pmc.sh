#!/bin/bash
# PMC passes (separate runs, kernel-trace only -- never with sys/hip traces) for the bench step.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/pmc
rocprofv3 -L > gpurun_out/pmc/counters_list.txt 2>&1
CMD="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-also --prewarm-seconds 0.3 ${WL:+--workload $WL}"
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
