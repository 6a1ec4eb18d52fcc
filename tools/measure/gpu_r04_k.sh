#!/bin/bash
# Round 4: up to which batch do the merged launches pay?  merge_max_batch 1000 (always) vs 0 (never) at B = 16, 24, 32 (whole step)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04k; mkdir -p $O
export SMK_GRAPH=1
timeout 900 python tools/measure/gpu_knob_ab.py merge_max_batch 10,16,24,32 1000,0 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/ab_merge.txt
