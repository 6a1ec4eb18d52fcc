#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r03o
O=$R/gpurun_out/r03o
export SMK_GRAPH=1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 200 rocprofv3 --kernel-trace -f csv -d $O/prof -- python $R/tools/measure/gpu_step_only.py 8 > $O/out.txt 2> $O/err.txt
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last 60 dispatches
t0=None
out=[]
for r in rows[-75:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if t0 is None: t0=s
    out.append("%9.2f %8.2f  q%-3s %s" % ((s-t0)/1e3,(e-s)/1e3,r.get("Queue_Id","?"),r["Kernel_Name"][:70]))
open("$O/trace_tail.txt","w").write("\n".join(out))
print("\n".join(out))
PY
find $O/prof -name "*kernel_trace.csv" -delete
