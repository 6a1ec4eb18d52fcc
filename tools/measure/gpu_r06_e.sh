#!/bin/bash
# Round 6: the matrix pipes' ceiling on random fp16 operands with NO memory traffic at all (conv_pp_kernel, MFMAs + barriers only, operand registers zero vs random)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06e; rm -rf $O; mkdir -p $O
SMK_LIB=$R/build_variants/measure/siammask_amd/libsiammask_hip.so timeout 900 python tools/measure/gpu_pp_ablate.py 64 20 3,67,3,67,66,2,0 2>&1 | grep -v amdgpu.ids | tee $O/pp_mfma_ceiling.txt
