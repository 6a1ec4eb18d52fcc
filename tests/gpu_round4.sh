#!/bin/bash
# parity + A/B of the launch-merging / streaming-store knobs + rocprofv3 kernel trace of the bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q --tb=short 2>&1 | grep -v amdgpu.ids | tail -6
for T in "merge=1" "merge=0" "merge=1,nt_store=0"; do
  timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-also --tune "$T" 2>/dev/null > gpurun_out/ab_$T.json
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$T.json")); print("$T", d["value"], "fps", d["ms_per_step"], "ms/step")
PY
done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also > $R/gpurun_out/rocprof_bench.json 2> $R/gpurun_out/rocprof.err
echo "rocprof exit $?"
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof/**/*kernel_trace.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # last full step: find the last cvt_in and print the sequence after it
    idx = [i for i, r in enumerate(rows) if "cvt_in" in r["Kernel_Name"]]
    a = idx[-2] if len(idx) > 1 else 0
    b = idx[-1] if len(idx) > 1 else len(rows)
    t0 = int(rows[a]["Start_Timestamp"])
    out = []
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        out.append("%8.1f %7.1f  %s grid=%s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:60], r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
    open("gpurun_out/step_timeline.txt", "w").write("\n".join(out))
    print("\n".join(out[:90]))
PY
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
