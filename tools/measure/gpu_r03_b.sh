#!/bin/bash
# Round 3, run B: the unchanged tools on the MI355X (oracle/_ref), probe 2 of the sequence K loop, the full GPU suite.
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r03b
O=gpurun_out/r03b
export SMK_GRAPH=1
timeout 1500 python -m pytest tests/test_gpu_tools.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest_tools.txt
timeout 600 python tools/measure/gpu_seq_probe2.py > $O/seq_probe2.txt 2>&1
timeout 1200 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_tools.py 2>&1 | tail -15 > $O/pytest_gpu.txt
timeout 400 python bench.py --steps 100 --warmup 10 > $O/bench.txt 2>&1
tail -n 3 $O/pytest_tools.txt $O/pytest_gpu.txt; tail -c 900 $O/bench.txt
