#!/bin/bash
# Round 3, run E: producers keep the layer geometry in (opaque) SGPRs, SeqLayer pairs dword aligned; 128x64 tile fixed
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r03e
O=gpurun_out/r03e
export SMK_GRAPH=1
timeout 600 python -m pytest tests/test_gpu_seq.py tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -5 > $O/pytest_seq_ops.txt
timeout 300 python tools/measure/gpu_knob_ab.py seq_first_stage 8 1,0 > $O/ab_first_stage.txt 2>&1
SMK_SEQ_CLK=2 timeout 300 python tools/measure/gpu_seqclk.py > $O/seqclk2.txt 2>&1
timeout 400 python bench.py --steps 100 --warmup 10 > $O/bench.txt 2>&1
tail -n 3 $O/pytest_seq_ops.txt; tail -n 1 $O/ab_first_stage.txt; head -c 400 $O/bench.txt | tail -c 330
