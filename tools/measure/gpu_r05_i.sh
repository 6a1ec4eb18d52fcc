#!/bin/bash
# Round 5 probe: how long is the front || tail region when the tail is LIGHT (mask head back on the main path: smk_tune chain_mask = 0)?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05i; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
cd /tmp && export TMPDIR=/tmp
for t in chain_mask=0 chain_mask=1; do
timeout 200 rocprofv3 --kernel-trace -f csv -d $O/prof_$t -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --prewarm-seconds 0.3 --no-cpu-baseline --no-also --no-long --tune $t > $O/rocprof_bench_$t.json 2> $O/rocprof.err
python - <<PY
import csv, glob, json
f = glob.glob("$O/prof_$t/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "stem_pool" in r["Kernel_Name"]]
k = 5 + 5 + 12
a, b = idx[k], idx[k + 2]
t0 = int(rows[a]["Start_Timestamp"])
print("---- $t", json.loads(open("$O/rocprof_bench_$t.json").read().strip().splitlines()[-1])["ms_per_step"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].replace("_ZN3smk", "").split("(")[0]
    print("%8.1f -> %8.1f dur %6.1f  q=%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), nm[:70]))
print("two steps span %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
PY
rm -rf $O/prof_$t
done
