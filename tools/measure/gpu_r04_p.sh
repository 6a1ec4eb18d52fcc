#!/bin/bash
# Round 4: smk_tune "seq_spoll" (team barrier polled through the scalar memory path) as a run-time flag: same-process ABAB on the
# fused step (B = 8, 16, 24, 5), the sequence tests with the flag on, the bench line off / on.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04p; mkdir -p $O
export SMK_GRAPH=1
SMK_TUNE=seq_spoll=1 timeout 600 python -m pytest tests/test_gpu_seq.py tests/test_gpu_e2e.py -x -q 2>&1 | grep -E "passed|failed" | tail -2 | tee -a $O/pytest.txt
timeout 600 python tools/measure/gpu_knob_ab.py seq_spoll 8,16,24,5 2>&1 | grep "ms/step" | tee -a $O/knob_ab.txt
timeout 300 python tools/measure/gpu_knob_ab.py seq_spoll 8 2>&1 | grep "ms/step" | tee -a $O/knob_ab.txt
B="python3 bench.py --steps 300 --warmup 20 --no-also --no-cpu-baseline --no-long"
for rep in 1 2; do
  for arm in on off; do
    unset SMK_TUNE; [ $arm = on ] && export SMK_TUNE=seq_spoll=1
    timeout 120 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm', d['value'], d['ms_per_step'], 'seq', d['roofline']['avg_launch_us'])" | tee -a $O/ab.txt
  done
done
