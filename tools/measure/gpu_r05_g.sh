#!/bin/bash
# Round 5: the tail's start as a gate kernel (pipe_sig = 2) against the event (0): parity + A/B + timeline
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05g; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ring.py tests/test_gpu_tracker.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -5 | tee $O/pytest_sig2.txt
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --workload sharp_b64_f16 > $O/b64.json 2>>$O/bench.err; timeout 300 python bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline --no-also --workload sharp_b1_f16 > $O/b1.json 2>>$O/bench.err; python -c "import json;[print(n, (lambda d:(d[\"value\"], d[\"ms_per_step\"], d[\"serial_steps\"][\"ms_per_step\"], d[\"latency\"]))(json.loads(open(\"$O/%s.json\"%n).read().strip().splitlines()[-1]))) for n in (\"b64\",\"b1\")]" 2>&1 | tail -3 | tee $O/pytest_seq.txt
B="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also"
for t in pipe_sig=0 pipe_sig=2 pipe_sig=0 pipe_sig=2; do
  timeout 300 python bench.py $B --tune $t > $O/bench_$t.json 2>> $O/bench.err
  python - <<PY
import json
d = json.loads(open("$O/bench_$t.json").read().strip().splitlines()[-1])
print("$t", d["value"], d["ms_per_step"], "200:", d.get("value_200_steps"), "lat:", d.get("latency"), "serial:", (d.get("serial_steps") or {}).get("ms_per_step"), d["config"]["persistent_sequences"])
PY
done
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
open("$O/pipelined_timeline_folded.txt", "w").write("\n".join(out))
print("\n".join(out))
PY
rm -rf $O/prof
