#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q --tb=short -k "chain or tight or fused_step or b8" 2>&1 | grep -v amdgpu.ids | tail -5
cat gpurun_out/e2e_refine_chain.json
SMK_CHAIN_CLK=1 timeout 300 python tools/measure/gpu_chain_prof.py 2>&1 | grep -v amdgpu.ids > gpurun_out/chain_prof.txt
grep -n "chain=1" gpurun_out/chain_prof.txt; grep "refine_chain layers" gpurun_out/chain_prof.txt | awk 'NR%8==0' | tail -6
AB_E2E_ONLY=1 timeout 600 python tools/measure/gpu_ab.py chain 0,1 2>&1 | grep -v amdgpu.ids | grep e2e | tee gpurun_out/chain_ab.txt
