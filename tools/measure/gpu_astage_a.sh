#!/bin/bash
# round 2, run A (short): what does the access pattern of the activation producers cost (tools/dma_patterns.hip), is the
# register-staged variant (smk_tune a_stage) bit-identical, and what does it do to the layers and to the whole step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
export LD_LIBRARY_PATH=/opt/rocm/lib:$LD_LIBRARY_PATH
timeout 90 ./tools/dma_patterns.bin > gpurun_out/dma_patterns.txt 2>&1; echo "dma_patterns exit $?"
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -m gpu -x -q -k "register_staged" 2>&1 | grep -v amdgpu.ids | tail -5
timeout 200 python tools/measure/gpu_knob_ab.py a_stage 8 > gpurun_out/astage_ab.txt 2>&1; echo "knob ab exit $?"
timeout 200 python tools/measure/gpu_producer_bench.py 8 gpurun_out/astage_bench.json > gpurun_out/astage_bench.txt 2>&1; echo "astage bench exit $?"
SMK_SEQ_CLK=1 SMK_TUNE=a_stage=0 timeout 100 python tools/measure/gpu_seqclk.py > gpurun_out/seqclk_a0.txt 2>&1
SMK_SEQ_CLK=1 SMK_TUNE=a_stage=1 timeout 100 python tools/measure/gpu_seqclk.py > gpurun_out/seqclk_a1.txt 2>&1
tail -40 gpurun_out/dma_patterns.txt; cat gpurun_out/astage_ab.txt | grep -v amdgpu.ids; grep -v amdgpu.ids gpurun_out/astage_bench.txt
