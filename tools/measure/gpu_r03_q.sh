#!/bin/bash
# Round 3, run Q: fused layer1 Bottleneck (l1_block_kernel): parity through the e2e suite, A/B on the step
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r03q
O=gpurun_out/r03q
export SMK_GRAPH=1
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -8 > $O/pytest_e2e.txt
timeout 300 python tools/measure/gpu_knob_ab.py l1_fused 8,1,64 0,1 > $O/ab_l1_fused.txt 2>&1
tail -n 6 $O/pytest_e2e.txt; grep ms/step $O/ab_l1_fused.txt
