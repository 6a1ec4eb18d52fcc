#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q --tb=short -k "chain or tight or fused_step or b8" 2>&1 | grep -v amdgpu.ids | tail -15
cat gpurun_out/e2e_refine_chain.json
timeout 300 python tests/gpu_chain_prof.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/chain_prof.txt
AB_E2E_ONLY=1 timeout 600 python tests/gpu_ab.py chain 0,1 2>&1 | grep -v amdgpu.ids | grep e2e | tee gpurun_out/chain_ab.txt
