#!/bin/bash
# fallback paths under SMK_TUNE + the default bench line with its layer profile
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
SMK_TUNE=chain=0,halo=0 timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_tracker.py -m gpu -x -q --tb=short 2>&1 | grep -v amdgpu.ids | tail -4
timeout 600 python bench.py --profile-out gpurun_out/bench_layers.json > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?"
cat gpurun_out/bench.json
