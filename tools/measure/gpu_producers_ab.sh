#!/bin/bash
# round 2: producer variants of conv_wreg / conv_seq (smk_tune a_stage, npw): bit-equality tests, whole-step A/B, per-layer
# A/B of one knob (AB_KNOB / AB_VALS, default npw 2,4), per-layer clocks inside the sequences
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
KNOB=${AB_KNOB:-npw}; VALS=${AB_VALS:-2,4}; V1=${VALS#*,}
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -x -q -k "producer_variants" 2>&1 | grep -v amdgpu.ids | tail -5
timeout 200 python tools/measure/gpu_knob_ab.py $KNOB 8 $VALS > gpurun_out/${KNOB}_ab.txt 2>&1; echo "knob ab exit $?"
AB_KNOB=$KNOB AB_VALS=$VALS timeout 200 python tools/measure/gpu_producer_bench.py 8 gpurun_out/${KNOB}_bench.json > gpurun_out/${KNOB}_bench.txt 2>&1; echo "layer bench exit $?"
SMK_SEQ_CLK=1 SMK_TUNE=$KNOB=$V1 timeout 100 python tools/measure/gpu_seqclk.py > gpurun_out/seqclk_${KNOB}${V1}.txt 2>&1
grep -v amdgpu.ids gpurun_out/${KNOB}_ab.txt; grep -v amdgpu.ids gpurun_out/${KNOB}_bench.txt | cut -c1-200
