#!/bin/bash
# Round 4: K-loop ablation of conv_wreg_kernel's 64x64 tile (MEASURE=1 library built into build_variants/measure_src)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04v; mkdir -p $O
SMK_LIB=$R/build_variants/measure_src/libsiammask_hip.so timeout 600 python tools/measure/gpu_ablate64.py 2>&1 | grep -v amdgpu.ids | tee $O/ablate64.txt
