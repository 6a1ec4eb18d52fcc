#!/bin/bash
# Round 4: the 64-row form of the per-launch pair (c3c1s_tile, pair_launch = 3): parity, then A/B against the two launches
# (pair_launch 0 vs 3) and against the 32-row form (2 vs 3) at B = 64, 32, 10, 1.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04i; mkdir -p $O
export SMK_GRAPH=1
timeout 600 python -m pytest tests/test_gpu_corr_head.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -8 | tee $O/pytest.txt
timeout 900 python tools/measure/gpu_knob_ab.py pair_launch 64,32,10,1 0,3 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/ab_pair64_vs_two.txt
timeout 600 python tools/measure/gpu_knob_ab.py pair_launch 32,10 2,3 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/ab_pair64_vs_pair32.txt
