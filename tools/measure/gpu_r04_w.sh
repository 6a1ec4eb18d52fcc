#!/bin/bash
# Round 4: triples -- [conv2, conv3, next 1x1] as one tile routine (c3c1_tile.inc FRONT = 1, smk_tune seq_fuse3): parity on first contact,
# then the step with it off | on, then the phase stamps.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04w; mkdir -p $O
export SMK_GRAPH=1
timeout 600 python -m pytest tests/test_gpu_seq.py -x -q -k "triples" 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 | tee $O/pytest_triples.txt
timeout 400 python tools/measure/gpu_knob_ab.py seq_fuse3 8,16 0,1 2>&1 | grep "ms/step" | tee $O/knob_ab.txt
timeout 200 python tools/measure/gpu_knob_ab.py seq_fuse3 8 0,2 2>&1 | grep "ms/step" | tee -a $O/knob_ab.txt
SMK_TUNE=seq_fuse3=1 SMK_GRAPH=0 SMK_SEQ_CLK=2 timeout 120 python tools/measure/gpu_seqclk.py 2>&1 | grep "seq clk\|seq arrive" | tail -75 > $O/stamps_on.txt
grep "total\|cfg 28\|cfg 29\|3x3 + conv3" $O/stamps_on.txt | cut -c1-330
SMK_TUNE=seq_fuse3=1 timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_e2e.txt
