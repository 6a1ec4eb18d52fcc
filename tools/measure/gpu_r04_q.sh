#!/bin/bash
# Round 4: Refine's front on the register-fed kernel (smk_tune "rf_wreg": bit 0 the merged window convolutions + deconv, bits 4..6 its
# tile code (0 = 64x64, 6 = 128x64); bit 1 the merged v*.2 launch, bits 8..10 its tile code).  Per-launch tables + same-process ABAB at B = 8.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04q; mkdir -p $O
export SMK_GRAPH=1
B="python3 bench.py --steps 200 --warmup 20 --no-also --no-cpu-baseline --no-long"
for v in 0 3 $((3+6*16)) $((3+6*16+6*256)) $((3+6*256)) $((3+2*16)) $((3+5*16)); do
  SMK_TUNE=rf_wreg=$v timeout 120 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rf_wreg=$v', d['value'], d['ms_per_step'], [(r['kernel'][:34], r['us_per_step']) for r in d['roofline']['kernels'] if 'merged' in r['kernel']])" | tee -a $O/tables2.txt
done
for v in 3 $((3+6*16)); do
  timeout 300 python tools/measure/gpu_knob_ab.py rf_wreg 8 0,$v 2>&1 | grep "ms/step" | tee -a $O/knob_ab2.txt
done
