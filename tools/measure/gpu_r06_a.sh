#!/bin/bash
# Round 6, first contact of conv_pp_kernel: parity (oracle + bit-equal to conv_wreg), then the per-layer A/B at B = 64
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06a; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "conv_pp" 2>&1 | tail -15 | tee $O/pytest.txt
timeout 600 python tools/measure/gpu_pp_bench.py 64 20 3 2>&1 | tee $O/pp_bench_b64.txt
timeout 300 python tools/measure/gpu_pp_bench.py 32 20 3 2>&1 | tee $O/pp_bench_b32.txt
