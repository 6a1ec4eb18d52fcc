#!/bin/bash
# Round 3, second half: with the faster sequence (fused pairs, patch-sharing tiles) -- for which batches does it pay now, and does the
# pair fusion pay where a team owns two / three images?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/sweep; mkdir -p $O
export SMK_GRAPH=1
timeout 400 python tools/measure/gpu_seq_batch_sweep.py 3,4,5,10,12,32 2>&1 | grep -v amdgpu.ids | tee $O/sweep.txt
timeout 200 python tools/measure/gpu_knob_ab.py seq_fuse 16,24 1,3 2>&1 | grep -v amdgpu.ids | tee $O/ab_fuse_b16_b24.txt
