#!/bin/bash
# Round 3, run C: K-loop skeleton (wave split sweep), the unchanged tools on the MI355X.
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r03c
O=gpurun_out/r03c
export SMK_GRAPH=1
timeout 300 ./tools/kloop_skeleton.bin > $O/kloop_skeleton.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_tools.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest_tools.txt
cp gpurun_out/tools_on_mi355x_*.json $O/ 2>/dev/null
tail -n 4 $O/pytest_tools.txt; cat $O/kloop_skeleton.txt
