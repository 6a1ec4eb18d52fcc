#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r03j
O=$R/gpurun_out/r03j
export SMK_GRAPH=1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -- python $R/tools/measure/gpu_step_only.py 8 > $O/out.txt 2> $O/err.txt
find $O/prof -name "*kernel_trace.csv" -delete
cd $R
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_step_only_b8.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats_step_only_b8.csv")))
for r in rows[:20]:
    print("%-84s calls %6s avg %9.2f us  %5s %%" % (r["Name"][:84], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
