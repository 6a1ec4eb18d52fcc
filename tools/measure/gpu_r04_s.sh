#!/bin/bash
# Round 4: conv_wreg_kernel's 64x64 tile with more of both operand streams in flight (smk_tune "wreg_deep": 1 = A ring 5 / weights 4
# K tiles ahead, 2 = 7 / 6; default 3 / 2).  In the real step the weights of the 64x64 launches (Refine's front at B = 8; 28 launches
# at B = 1) come from beyond the L2 every frame -- the per-layer loops of round 2 that rejected a deeper ring kept them warm.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04s; mkdir -p $O
export SMK_GRAPH=1
for v in 1 2; do
  timeout 400 python tools/measure/gpu_knob_ab.py wreg_deep 8,1,2,16 0,$v 2>&1 | grep "ms/step" | tee -a $O/knob_ab.txt
done
B="python3 bench.py --steps 200 --warmup 20 --no-also --no-cpu-baseline --no-long"
for v in 0 1 2; do
  SMK_TUNE=wreg_deep=$v timeout 120 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wreg_deep=$v', d['value'], d['ms_per_step'], [(r['kernel'][:34], r['us_per_step']) for r in d['roofline']['kernels'] if 'merged' in r['kernel']])" | tee -a $O/tables.txt
done
for v in 1 2; do
  SMK_TUNE=wreg_deep=$v,wreg=4 timeout 600 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -2 | tee -a $O/pytest.txt
done
SMK_TUNE=wreg_deep=1 timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_corr_head.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -2 | tee -a $O/pytest.txt
