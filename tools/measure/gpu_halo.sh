#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q --tb=short -k halo 2>&1 | grep -v amdgpu.ids | tail -15
SMK_TUNE=halo=1 timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q --tb=short 2>&1 | grep -v amdgpu.ids | tail -8
timeout 900 python tools/measure/gpu_halo.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/halo_ab.txt
