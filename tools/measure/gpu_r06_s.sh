#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06s; rm -rf $O; mkdir -p $O
export SMK_LIB=$R/build_variants/measure/siammask_amd/libsiammask_hip.so
ls -la $SMK_LIB
timeout 600 python tools/measure/gpu_wreg_narrow_ablate.py 8 2>&1 | tee $O/ablate_b8.txt
