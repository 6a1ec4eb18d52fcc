#!/bin/bash
# Round 4: does carrying the triple routine in conv_seq_kernel cost the default path anything (196 against 100 spilled SGPRs)?
# product library (seq_fuse3 = 0) against a variant compiled without the routine, alternating processes.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04x; mkdir -p $O
export SMK_GRAPH=1
B="python3 bench.py --steps 400 --warmup 20 --no-also --no-cpu-baseline --no-long"
for rep in 1 2 3 4; do
  for arm in product notriple; do
    unset SMK_LIB; [ $arm != product ] && export SMK_LIB=$R/build_variants/$arm/libsiammask_hip.so
    timeout 120 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm', d['value'], d['ms_per_step'], 'seq', d['roofline']['avg_launch_us'])" | tee -a $O/ab.txt
  done
done
