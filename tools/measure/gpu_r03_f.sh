#!/bin/bash
# Round 3, run F: full-height dw_xcorr (parity, A/B), igemm producer pinning (whole suite), per-layer profile of the step
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r03f
O=gpurun_out/r03f
export SMK_GRAPH=1
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > $O/pytest_gpu.txt
timeout 300 python tools/measure/gpu_knob_ab.py xc_full 8,1,64 0,1 > $O/ab_xc_full.txt 2>&1
timeout 400 python bench.py --steps 100 --warmup 10 --profile-out $O/layers_b8.json > $O/bench.txt 2>&1
tail -n 3 $O/pytest_gpu.txt; grep ms/step $O/ab_xc_full.txt; head -c 400 $O/bench.txt | tail -c 330
