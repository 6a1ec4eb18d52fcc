#!/bin/bash
# Round 4: conv_pair_kernel (conv3 + next 1x1 as one launch outside the persistent sequence): parity, then the knob A/B at
# B = 1, 64, 32, 4, 10 (one process each, off/on/off/on).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04h; mkdir -p $O
export SMK_GRAPH=1
timeout 600 python -m pytest tests/test_gpu_corr_head.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -8 | tee $O/pytest_pair_launch.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -6 | tee $O/pytest_e2e.txt
timeout 600 python tools/measure/gpu_knob_ab.py pair_launch 1,64,32,4,10 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/ab_pair_launch.txt
