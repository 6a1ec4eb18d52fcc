#!/bin/bash
# Round 5, first contact of the software-pipelined frame step (smk_set_pipeline): parity test, bench A/B (pipelined vs serial,
# graph vs eager halves), kernel-trace timeline of pipelined steps.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05b; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ring.py -x -q 2>&1 | tail -15 | tee $O/pytest_pipeline.txt
B="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also"
timeout 300 python bench.py $B > $O/bench_pipe.json 2> $O/bench_pipe.err; echo "bench exit $?"
for e in pipe_eager=2 pipe_join=0 pipe_join=0,pipe_eager=2; do
  timeout 200 python bench.py $B --no-long --tune $e > $O/bench_pipe_$e.json 2>> $O/bench_pipe.err
done
timeout 200 python bench.py $B --workload sharp_b64_f16 --steps 20 > $O/bench_b64_pipe.json 2>> $O/bench_pipe.err
timeout 200 python bench.py $B --workload sharp_b1_f16 --steps 50 > $O/bench_b1_pipe.json 2>> $O/bench_pipe.err
python - <<PY
import json
for n in ("bench_pipe", "bench_pipe_pipe_eager=2", "bench_pipe_pipe_join=0", "bench_pipe_pipe_join=0,pipe_eager=2", "bench_b64_pipe", "bench_b1_pipe"):
    try:
        d = json.loads(open("$O/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], "200:", d.get("value_200_steps"), "lat:", d.get("latency"), "serial:", d.get("serial_steps"), d["timing"])
    except Exception as e:
        print(n, "ERR", e)
PY
tail -5 $O/bench_pipe.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -f csv -d $O/prof -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --prewarm-seconds 0.3 --no-cpu-baseline --no-also --no-long > $O/rocprof_bench.json 2> $O/rocprof.err
echo "rocprof exit $?"
cd $R
python - <<PY
import csv, glob
f = glob.glob("$O/prof/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "stem_pool" in r["Kernel_Name"]]
print(len(idx), "steps in trace")
k = 5 + 5 + 12                 # a pipelined timed step (prewarm/warm-up are in front, serial + profile passes behind)
a, b = idx[k], idx[k + 2]
t0 = int(rows[a]["Start_Timestamp"])
out = []
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].replace("_ZN3smk", "").split("(")[0]
    out.append("%8.1f -> %8.1f dur %6.1f  q=%s %s grid=%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), nm[:70], r.get("Grid_Size_X", "?")))
out.append("two steps span %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
open("$O/pipelined_timeline.txt", "w").write("\n".join(out))
print("\n".join(out))
PY
rm -rf $O/prof
