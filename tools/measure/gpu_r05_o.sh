#!/bin/bash
# Round 5: depth-2 pipelining, form 2 (chain + mask head launch gated on corr_head's START, i.e. behind conv_search): parity + A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05o; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
SMK_TUNE=pipe_two_form=2 timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_corr_head.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
B="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also"
for a in "1 pipe_two_form=1" "2 pipe_two_form=2" "2 pipe_two_form=1" "1 pipe_two_form=1" "2 pipe_two_form=2" "2 pipe_two_form=1"; do
  set -- $a
  timeout 300 python bench.py $B --pipeline-depth $1 --tune $2 > $O/b.json 2>> $O/bench.err
  python - <<PY
import json
d = json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print("depth $1 $2", d["value"], d["ms_per_step"], "200:", d["value_200_steps"]["ms_per_step"], "serial:", d["serial_steps"]["ms_per_step"], "lat", d["latency"]["box_ms_median"], d["latency"]["mask_ms_median"], d["config"]["persistent_sequences"]["err"])
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -f csv -d $O/prof -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --prewarm-seconds 0.3 --no-cpu-baseline --no-also --no-long --pipeline-depth 2 --tune pipe_two_form=2 > $O/rocprof_bench.json 2> $O/rocprof.err
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
open("$O/timeline_form2.txt", "w").write("\n".join(out))
print("\n".join(out))
PY
rm -rf $O/prof
