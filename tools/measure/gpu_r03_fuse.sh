#!/bin/bash
# Round 3: first contact of the fused (conv3, next 1x1) tile routine (c3c1_tile.inc): its per-op tests, where it differs if it
# does, the B = 8 end-to-end gates, the A/B of the step in one process and the per-layer stamps with the pairs fused.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/fuse; mkdir -p $O
export SMK_GRAPH=1
timeout 120 python tools/measure/gpu_fuse_debug.py > $O/debug.txt 2>&1; echo "debug exit $?"; cat $O/debug.txt | tail -30
timeout 400 python -m pytest tests/test_gpu_seq.py -q 2>&1 | grep -v amdgpu.ids | tail -40 > $O/pytest_seq.txt; tail -15 $O/pytest_seq.txt
timeout 400 python -m pytest tests/test_gpu_e2e.py -q -k "b8 or persistent or which_batches or producer_variants" 2>&1 | grep -v amdgpu.ids | tail -40 > $O/pytest_e2e.txt; tail -12 $O/pytest_e2e.txt
timeout 300 python tools/measure/gpu_knob_ab.py seq_fuse 8 0,1 > $O/ab_seq_fuse.txt 2>&1; cat $O/ab_seq_fuse.txt
timeout 200 python tools/measure/gpu_knob_ab.py seq_halo 8,16 0,1 > $O/ab_seq_halo.txt 2>&1; cat $O/ab_seq_halo.txt
SMK_GRAPH=0 SMK_SEQ_CLK=2 timeout 120 python tools/measure/gpu_seqclk.py > $O/seqclk.txt 2>&1; grep "seq clk" $O/seqclk.txt | tail -70
