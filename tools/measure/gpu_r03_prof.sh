#!/bin/bash
# rocprofv3 kernel stats of the driver's bench command (graph replay), per-kernel average durations
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r03prof
O=$R/gpurun_out/r03prof
export SMK_GRAPH=1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -- python $R/bench.py --steps 20 --warmup 5 --prewarm-seconds 0.3 --no-cpu-baseline --no-also > $O/rocprof_bench.json 2> $O/rocprof.err
echo "rocprof exit $?"
find $O/prof -name "*kernel_trace.csv" -delete
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $O/kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
for r in rows[:22]:
    print("%-90s calls %6s avg %9.2f us  %5s %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
