#!/bin/bash
# rocprofv3 kernel trace of a few graph-replayed steps: per-kernel durations and inter-kernel gaps
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tl
timeout 300 rocprofv3 --kernel-trace -f csv -d $R/gpurun_out/prof_tl -- python $R/bench.py --steps 30 --warmup 5 --prewarm-seconds 0.5 --no-cpu-baseline --no-also ${WL:+--workload $WL} > $R/gpurun_out/tl_bench.json 2> $R/gpurun_out/tl.err
echo "rocprof exit $?"
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_tl/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "cvt_in" in r["Kernel_Name"]]
print(len(idx), "steps in trace")
k = len(idx) - 12          # a graph-replayed timed step (the last 3 are the eager profile pass)
a, b = idx[k], idx[k + 1]
t0 = int(rows[a]["Start_Timestamp"])
out, tot, gaps, pe = [], 0, 0, None
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = 0 if pe is None else s - pe
    tot += e - s; gaps += max(0, g); pe = max(pe or 0, e)
    nm = r["Kernel_Name"].replace("_ZN3smk17conv_igemm_kernelI", "igemm<").replace("EEEvNS_9ConvBatchE", ">")
    out.append("%8.1f dur %6.1f gap %5.1f  %s wg=%s" % ((s - t0) / 1e3, (e - s) / 1e3, g / 1e3, nm[:60], r.get("Workgroup_Size_X", "?") + "x" + r.get("Grid_Size_X", "?")))
span = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
out.append("kernels %d  sum_dur %.1f us  sum_gaps %.1f us  step span %.1f us" % (b - a, tot / 1e3, gaps / 1e3, span))
open("gpurun_out/step_timeline.txt", "w").write("\n".join(out))
print("\n".join(out))
PY
rm -rf gpurun_out/prof_tl
