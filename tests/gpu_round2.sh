#!/bin/bash
# parity tests + per-layer conv micro-benchmark + a short end-to-end bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q --tb=short > gpurun_out/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.log
tail -15 gpurun_out/pytest.log
timeout 900 python tests/gpu_convbench.py ${BATCHES:-8} gpurun_out/convbench.json f16 > gpurun_out/convbench.log 2>&1; echo "convbench exit $?" >> gpurun_out/convbench.log
tail -40 gpurun_out/convbench.log
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --profile-out gpurun_out/bench_layers.json > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit $?" >> gpurun_out/bench.err
cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
