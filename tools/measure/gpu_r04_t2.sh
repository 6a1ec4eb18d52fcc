#!/bin/bash
# Round 4: rf_wreg 3 | 7 | 15 at B = 8, more alternations (the step-level effect of a 5-8 us change is inside one run's noise)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04t; mkdir -p $O
export SMK_GRAPH=1
python - <<'PY' | tee -a $O/long_ab.txt
import os, sys
sys.path.insert(0, "tools/measure")
from gpu_seq_ab import run
import statistics
res = {3: [], 7: [], 15: []}
for rep in range(6):
    for v in (3, 7, 15):
        d, st, o = run(8, 1, steps=500, extra={"rf_wreg": v})
        res[v].append(d)
for v in res:
    print("rf_wreg=%d  ms/step: %s   median %.4f  mean %.4f" % (v, " ".join("%.4f" % x for x in res[v]), statistics.median(res[v]), statistics.mean(res[v])), flush=True)
PY
