#!/bin/bash
# Round 4: how much of a team wait is skew?  Arrival stamps of all 32 workgroups of team 0 at every barrier (SMK_SEQ_CLK=2 build).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04m; mkdir -p $O
SMK_GRAPH=0 SMK_SEQ_CLK=2 timeout 120 python tools/measure/gpu_seqclk.py 2>&1 | grep "seq arrive\|total" > $O/arrivals.txt
tail -40 $O/arrivals.txt
