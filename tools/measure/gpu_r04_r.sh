#!/bin/bash
# Round 4: rf_wreg = 3 as the default (Refine's two merged front launches on the register-fed kernel, 64x64): the other batches
# (same-process ABAB, 3 = new default against 0) and the GPU suite's Refine / end-to-end files on the new default.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04r; mkdir -p $O
export SMK_GRAPH=1
timeout 600 python tools/measure/gpu_knob_ab.py rf_wreg 1,2,5,12,16,24 0,3 2>&1 | grep "ms/step" | tee -a $O/knob_ab.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_ops.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest.txt
