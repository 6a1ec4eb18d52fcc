#!/bin/bash
# Round 4: long-K members of Refine's merged front launch on 32-row tiles (ConvParams::tile32; smk_tune rf_wreg bit 2: v2.0, bit 3: v1.0 too).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04t; mkdir -p $O
export SMK_GRAPH=1
B="python3 bench.py --steps 200 --warmup 20 --no-also --no-cpu-baseline --no-long"
for v in 3 7 15; do
  SMK_TUNE=rf_wreg=$v timeout 120 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rf_wreg=$v', d['value'], d['ms_per_step'], [(r['kernel'][:34], r['us_per_step']) for r in d['roofline']['kernels'] if 'merged' in r['kernel']])" | tee -a $O/tables.txt
done
for v in 7 15; do
  timeout 400 python tools/measure/gpu_knob_ab.py rf_wreg 8,1,16 3,$v 2>&1 | grep "ms/step" | tee -a $O/knob_ab.txt
done
