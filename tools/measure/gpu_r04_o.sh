#!/bin/bash
# Round 4: the team barrier's poll through the SCALAR memory path (s_load_dword glc) instead of a vector sc1 load: it then does not queue
# behind the workgroup's own vector requests (the fused pairs' residual rows).  Variant library (build_variant.sh spoll).
# Parity (bit-identical: only the poll changes), ABAB on the whole step, release stamps.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04o; mkdir -p $O
export SMK_GRAPH=1
SMK_LIB=$R/build_variants/spoll/libsiammask_hip.so timeout 600 python -m pytest tests/test_gpu_seq.py -x -q 2>&1 | grep -E "passed|failed" | tail -2 | tee -a $O/pytest.txt
B="python3 bench.py --steps 300 --warmup 20 --no-also --no-cpu-baseline --no-long"
for rep in 1 2 3; do
  for arm in product spoll; do
    unset SMK_LIB; [ $arm != product ] && export SMK_LIB=$R/build_variants/$arm/libsiammask_hip.so
    timeout 120 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm', d['value'], d['ms_per_step'], 'seq', d['roofline']['avg_launch_us'])" | tee -a $O/ab.txt
  done
done
unset SMK_LIB
for arm in product spoll; do
  [ $arm != product ] && export SMK_LIB=$R/build_variants/$arm/libsiammask_hip.so
  echo "== $arm" >> $O/stamps.txt
  SMK_GRAPH=0 SMK_SEQ_CLK=2 timeout 120 python tools/measure/gpu_seqclk.py 2>&1 | grep "seq arrive\|seq clk2\|total" >> $O/stamps.txt
  unset SMK_LIB
done
grep "l3.2\|total" $O/stamps.txt | cut -c1-330
