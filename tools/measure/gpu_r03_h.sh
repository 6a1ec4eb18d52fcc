#!/bin/bash
# Round 3, run H: fused stem (stem_pool_kernel): parity through the e2e suite, A/B on the step
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r03h
O=gpurun_out/r03h
export SMK_GRAPH=1
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_dropin.py tests/test_gpu_tracker.py -x -q -m gpu 2>&1 | tail -5 > $O/pytest_e2e.txt
timeout 300 python tools/measure/gpu_knob_ab.py stem_fused 8,1,64 0,1 > $O/ab_stem_fused.txt 2>&1
tail -n 3 $O/pytest_e2e.txt; grep ms/step $O/ab_stem_fused.txt
