#!/bin/bash
# Round 3, first hardware contact of the re-structured conv_seq_kernel (split barrier + hoisted prologue, one launch,
# K-loop stagger knob, deep-ring variant): parity of the new per-op entry, phase stamps, knob A/Bs, micro-benchmarks.
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out/r03a
O=gpurun_out/r03a
export SMK_GRAPH=1
timeout 600 python -m pytest tests/test_gpu_seq.py -x -q -m gpu 2>&1 | tail -15 > $O/pytest_seq.txt
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "bench_configuration_b8 or persistent_sequences or producer_variants" 2>&1 | tail -15 > $O/pytest_e2e_b8.txt
SMK_SEQ_CLK=2 timeout 300 python tools/measure/gpu_seqclk.py > $O/seqclk2.txt 2>&1
timeout 600 python tools/measure/gpu_seq_probe.py > $O/seq_probe.txt 2>&1
timeout 300 python tools/measure/gpu_knob_ab.py seq_kstag 8 0,1 > $O/ab_kstag01.txt 2>&1
timeout 300 python tools/measure/gpu_knob_ab.py seq_kstag 8 0,2 > $O/ab_kstag02.txt 2>&1
timeout 300 python tools/measure/gpu_knob_ab.py seq_deep 8 0,1 > $O/ab_deep.txt 2>&1
timeout 300 python bench.py --steps 100 --warmup 10 > $O/bench.txt 2>&1
tail -3 $O/pytest_seq.txt $O/pytest_e2e_b8.txt; tail -2 $O/ab_*.txt; tail -c 600 $O/bench.txt
