#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q --tb=short -k halo 2>&1 | grep -v amdgpu.ids | tail -6
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q --tb=short -k "fp16 or chain or b8 or fused" 2>&1 | grep -v amdgpu.ids | tail -4
AB_E2E_ONLY=1 timeout 600 python tools/measure/gpu_ab.py halo_db 0,1 2>&1 | grep -v amdgpu.ids | grep e2e | tee gpurun_out/halo_db_ab.txt
