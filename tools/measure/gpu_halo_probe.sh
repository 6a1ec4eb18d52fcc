#!/bin/bash
# Round 3: the patch-sharing tile with a weight ring of 3 instead of 6 k-steps (a second library, built with -DSMK_HALO_D128=3 into
# build_variants/): is its K loop latency-bound on the weight stream?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/haloprobe; mkdir -p $O
timeout 200 python tools/measure/gpu_halo_probe.py "ring 6:" 2>&1 | grep -v amdgpu.ids | tee $O/probe.txt
cp siammask_amd/libsiammask_hip.so /tmp/lib_product.so
cp build_variants/libsiammask_hip_halo_d3.so siammask_amd/libsiammask_hip.so
timeout 200 python tools/measure/gpu_halo_probe.py "ring 3:" 2>&1 | grep -v amdgpu.ids | tee -a $O/probe.txt
cp /tmp/lib_product.so siammask_amd/libsiammask_hip.so
