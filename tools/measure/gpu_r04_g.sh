#!/bin/bash
# Round 4: where do the __amd_rocclr_copyBuffer dispatches of a steady-state step come from?  Kernel + memory-copy trace of the
# bench loop, one step's timeline printed; then the step rate under the HIP runtime's graph / kernarg knobs.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04g; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -f csv -d $O/prof -- python $R/bench.py --steps 40 --warmup 5 --prewarm-seconds 0.2 --no-cpu-baseline --no-also --no-long > $O/trace_bench.json 2> $O/trace.err
cd $R
python - <<PY | tee $O/step_timeline.txt
import csv, glob
ev=[]
for f in glob.glob("$O/prof/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], "q%s" % r.get("Queue_Id","?")))
for f in glob.glob("$O/prof/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "MEMCPY %s %s B" % (r.get("Direction","?"), r.get("Bytes", r.get("Size","?"))), ""))
ev.sort()
# last 3 steps: find stem_pool occurrences
idx=[i for i,e in enumerate(ev) if "stem_pool" in e[2]]
print("events", len(ev), "steps seen", len(idx))
i0=idx[-4]
t0=ev[i0][0]
for e in ev[i0:idx[-2]+1]:
    print("%9.2f us  +%7.2f  %s %s" % ((e[0]-t0)/1e3, (e[1]-e[0])/1e3, e[2], e[3]))
PY
rm -rf $O/prof
B="python3 bench.py --steps 300 --warmup 20 --no-also --no-cpu-baseline --no-long"
for envs in "" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "SMK_GRAPH=0" ""; do
  env $envs timeout 120 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('env [$envs]', d['value'], d['ms_per_step'], 'host enqueue', d['timing']['host_enqueue_ms_per_step'], 'gpu event', d['timing']['gpu_event_ms_per_step'])" | tee -a $O/env_ab.txt
done
