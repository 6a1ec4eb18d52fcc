#!/bin/bash
# Round 4: the product instantiation of conv_seq_kernel without the routines the default lists never use (pair split, deep-ring and
# 128x64 tiles: 50 against 103 spilled SGPRs) against the library before that change; alternating processes; the sequence tests.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04ab; mkdir -p $O
export SMK_GRAPH=1
timeout 900 python -m pytest tests/test_gpu_seq.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_seq.txt
B="python3 bench.py --steps 400 --warmup 20 --no-also --no-cpu-baseline --no-long"
for rep in 1 2 3 4; do
  for arm in product seqprev; do
    unset SMK_LIB; [ $arm != product ] && export SMK_LIB=$R/build_variants/$arm/libsiammask_hip.so
    timeout 120 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm', d['value'], d['ms_per_step'], 'seq', d['roofline']['avg_launch_us'])" | tee -a $O/ab.txt
  done
done
