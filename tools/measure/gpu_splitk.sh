#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q --tb=short -k "split_k" 2>&1 | grep -v amdgpu.ids | tail -12
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_tracker.py -m gpu -x -q --tb=short 2>&1 | grep -v amdgpu.ids | tail -6
AB_E2E_ONLY=1 timeout 600 python tools/measure/gpu_ab.py ksplit 0,1 2>&1 | grep -v amdgpu.ids | grep e2e | tee gpurun_out/ksplit_ab.txt
