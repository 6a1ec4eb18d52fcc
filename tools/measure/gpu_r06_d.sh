#!/bin/bash
# Round 6: conv_pp_kernel, second schedule (reads before the stage, B0 one phase early: 4/8/4/8 reads per phase): parity, ablations, per-layer A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06d; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "conv_pp" 2>&1 | tail -5 | tee $O/pytest.txt
SMK_LIB=$R/build_variants/measure/siammask_amd/libsiammask_hip.so timeout 900 python tools/measure/gpu_pp_ablate.py 64 20 0,32,8,1,2,3,4,6,5,0 2>&1 | grep -v amdgpu.ids | tee $O/pp_ablate_b64.txt
timeout 600 python tools/measure/gpu_pp_bench.py 64 20 3 2>&1 | grep -v amdgpu.ids | tee $O/pp_bench_b64.txt
SMK_BENCH_FILL=relu timeout 600 python tools/measure/gpu_pp_bench.py 64 20 3 2>&1 | grep -v amdgpu.ids | tee $O/pp_bench_b64_relu.txt
