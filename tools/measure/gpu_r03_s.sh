#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r03s
export SMK_GRAPH=1
SMK_L1_CLK=1 timeout 300 python tools/measure/gpu_seqclk.py 2>&1 | grep "l1 clk" | tail -6
