#!/bin/bash
# Round 4: with Refine's merged front launches on the register-fed kernel, is "merge" still wrong beyond 24 streams?  merge=1 (rule: off
# beyond merge_max_batch = 24) against merge=2 (always) at B = 32 and 64, same process.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04z; mkdir -p $O
export SMK_GRAPH=1
timeout 600 python tools/measure/gpu_knob_ab.py merge 32,64 1,2 2>&1 | grep "ms/step" | tee $O/knob_ab.txt
timeout 600 python tools/measure/gpu_knob_ab.py merge_max_batch 32 24,32 2>&1 | grep "ms/step" | tee -a $O/knob_ab.txt
