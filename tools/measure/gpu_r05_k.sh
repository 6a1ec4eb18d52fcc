#!/bin/bash
# Round 5: depth-2 pipelining (Refine chain + mask head of frame f beside the heads of frame f + 1): parity + ABAB + timeline
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05k; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ring.py tests/test_gpu_tracker.py tests/test_gpu_dropin.py tests/test_gpu_dist.py -x -q 2>&1 | tail -8 | tee $O/pytest.txt
B="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also"
for d in 1 2 1 2; do
  timeout 300 python bench.py $B --pipeline-depth $d > $O/bench_d$d.json 2>> $O/bench.err
  python - <<PY
import json
d = json.loads(open("$O/bench_d$d.json").read().strip().splitlines()[-1])
print("depth $d", d["value"], d["ms_per_step"], "200:", d.get("value_200_steps"), "lat:", d.get("latency"), "serial:", (d.get("serial_steps") or {}).get("ms_per_step"), (d.get("serial_steps") or {}).get("depth1"), d["config"]["persistent_sequences"])
PY
done
timeout 300 python bench.py $B --workload sharp_b16_f16 --pipeline-depth 2 --no-long > $O/b16_d2.json 2>> $O/bench.err
timeout 300 python bench.py $B --workload sharp_b16_f16 --pipeline-depth 1 --no-long > $O/b16_d1.json 2>> $O/bench.err
python -c "
import json
for n in ('b16_d2','b16_d1'):
    d=json.loads(open('$O/%s.json'%n).read().strip().splitlines()[-1]); print(n, d['value'], d['ms_per_step'])"
tail -3 $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -f csv -d $O/prof -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --prewarm-seconds 0.3 --no-cpu-baseline --no-also --no-long > $O/rocprof_bench.json 2> $O/rocprof.err
cd $R
python - <<PY
import csv, glob
f = glob.glob("$O/prof/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "stem_pool" in r["Kernel_Name"]]
k = 5 + 5 + 12
a, b = idx[k], idx[k + 2]
t0 = int(rows[a]["Start_Timestamp"])
out = []
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].replace("_ZN3smk", "").split("(")[0]
    out.append("%8.1f -> %8.1f dur %6.1f  q=%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), nm[:70]))
out.append("two steps span %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
open("$O/pipelined_timeline_depth2.txt", "w").write("\n".join(out))
print("\n".join(out))
PY
rm -rf $O/prof
