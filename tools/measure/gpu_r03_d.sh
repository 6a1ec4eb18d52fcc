#!/bin/bash
# Round 3, run D: layer1 inside the persistent sequence (128x64 tiles), A/B + per-layer clocks
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r03d
O=gpurun_out/r03d
export SMK_GRAPH=1
timeout 600 python -m pytest tests/test_gpu_seq.py -x -q -m gpu -k "every_tile" 2>&1 | tail -5 > $O/pytest_seq.txt
timeout 300 python tools/measure/gpu_knob_ab.py seq_first_stage 8 1,0 > $O/ab_first_stage.txt 2>&1
SMK_TUNE=seq_first_stage=0 SMK_SEQ_CLK=1 timeout 300 python tools/measure/gpu_seqclk.py > $O/seqclk_l1.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_tools.py -x -q -m gpu 2>&1 | tail -4 > $O/pytest_tools.txt
tail -n 3 $O/pytest_seq.txt $O/pytest_tools.txt; tail -n 2 $O/ab_first_stage.txt; tail -n 50 $O/seqclk_l1.txt
