#!/bin/bash
# op parity + conv micro-benchmark at several batch sizes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q --tb=short 2>&1 | grep -v amdgpu.ids | tail -8
for B in ${BATCHES:-8 64 1}; do
  timeout 900 python tools/measure/gpu_convbench.py $B gpurun_out/convbench_b$B.json f16 2>&1 | grep -v amdgpu.ids > gpurun_out/convbench_b$B.log
  echo "B=$B exit $?"; tail -31 gpurun_out/convbench_b$B.log
done
