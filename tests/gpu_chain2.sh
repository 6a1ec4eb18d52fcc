#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
SMK_CHAIN_CLK=1 timeout 300 python tests/gpu_chain_prof.py 2>&1 | grep -v amdgpu.ids > gpurun_out/chain_prof.txt
grep "refine_chain layers" gpurun_out/chain_prof.txt | awk 'NR%16==15 || NR%16==0' | tail -8
