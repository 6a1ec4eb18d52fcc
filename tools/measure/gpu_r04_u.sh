#!/bin/bash
# Round 4: conv_search (3 x (256 -> 256, 3x3) as one N = 768 GEMM, M = 841 B) on other conv_wreg tiles: 128x256 makes 159 workgroups at
# B = 8 (97 CUs idle), 96x256 (new instantiation) 213.  smk_tune "cs_tile": 0 rule (128x256), 7 = 96x256, 1 = 64x256, 5 = 128x128.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04u; mkdir -p $O
export SMK_GRAPH=1
B="python3 bench.py --steps 200 --warmup 20 --no-also --no-cpu-baseline --no-long"
for v in 0 7 1 5; do
  SMK_TUNE=cs_tile=$v timeout 120 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cs_tile=$v', d['value'], d['ms_per_step'], [(r['kernel'][:34], r['us_per_step']) for r in d['roofline']['kernels'] if 'conv_wreg' in r['kernel'] and 'merged' not in r['kernel']])" | tee -a $O/tables.txt
done
python - <<'PY' | tee -a $O/long_ab.txt
import os, sys
sys.path.insert(0, "tools/measure")
from gpu_seq_ab import run
import statistics
for B, vals in ((8, (0, 7)), (5, (0, 7)), (12, (0, 7)), (16, (0, 7))):
    res = {v: [] for v in vals}
    outs = {}
    for rep in range(4):
        for v in vals:
            d, st, o = run(B, 1, steps=400, extra={"cs_tile": v})
            res[v].append(d); outs[v] = o
    err = {k: float((outs[vals[0]][k].double() - outs[vals[1]][k].double()).abs().max()) for k in outs[vals[0]]}
    for v in res:
        print("B=%d cs_tile=%d  ms/step: %s   median %.4f" % (B, v, " ".join("%.4f" % x for x in res[v]), statistics.median(res[v])), flush=True)
    print("   max|diff|", err, flush=True)
PY
