#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r03n
O=gpurun_out/r03n
export SMK_GRAPH=1
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k xcorr 2>&1 | tail -3 > $O/pytest_xcorr.txt
timeout 300 python tools/measure/gpu_knob_ab.py xc_full 8,64 0,1 > $O/ab_xc_full.txt 2>&1
timeout 300 python tools/measure/gpu_knob_ab.py res_nt 8,64 0,1 > $O/ab_res_nt.txt 2>&1
tail -n 2 $O/pytest_xcorr.txt; grep -h ms/step $O/ab_*.txt
