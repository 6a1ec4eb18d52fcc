#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r03m
O=gpurun_out/r03m
export SMK_GRAPH=1
timeout 300 python tools/measure/gpu_knob_ab.py ksplit 8 0,1 > $O/ab_ksplit.txt 2>&1
timeout 300 python tools/measure/gpu_knob_ab.py seq_tall 8 2,1 > $O/ab_seq_tall.txt 2>&1
timeout 300 python tools/measure/gpu_knob_ab.py chain_mask 8 1,0 > $O/ab_chain_mask.txt 2>&1
grep -h ms/step $O/ab_*.txt
