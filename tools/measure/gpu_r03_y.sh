#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r03y; mkdir -p $O
timeout 800 python tools/measure/gpu_seq_batch_sweep.py > $O/seq_batch_sweep.txt 2>&1; echo "sweep exit $?"
