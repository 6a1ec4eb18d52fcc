#!/bin/bash
# Round 4, fourth GPU contact: corr_head_kernel (dw_xcorr + head.0 + cls / loc head.3 as one launch): parity against the three
# launches and the oracle gates, then the knob A/B on the whole step at B = 8 / 1 / 64 (one process each: off/on/off/on).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04d; mkdir -p $O
export SMK_GRAPH=1
timeout 600 python -m pytest tests/test_gpu_corr_head.py -x -q 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/pytest_corr_head.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_dropin.py tests/test_gpu_tracker.py tests/test_gpu_ring.py -x -q 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/pytest_e2e.txt
timeout 400 python tools/measure/gpu_knob_ab.py corr_head 8,1,64 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/ab_corr_head.txt
timeout 120 python3 bench.py --steps 100 --warmup 10 --no-also --no-cpu-baseline --no-long 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], 'launches', d['roofline']['launches_per_step_all_kernels'])
for r in d['roofline']['kernels']: print('   ', r)" | tee $O/bench_b8.txt
