#!/bin/bash
# Round 4: the poller wave of c3c1_tile carries loads in flight at the hoist point -> its poll's answer queues behind them.  Variants
# (separate libraries, tools/measure/build_variant.sh): RES_LATE = 1 (residual rows requested behind the wait by every wave), 2 (only
# by the poller's wave).  Parity (bit-identical: only the time of a load changes), ABAB on the whole step, arrival / release stamps.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04n; mkdir -p $O
export SMK_GRAPH=1
for v in 1 2; do
  SMK_LIB=$R/build_variants/reslate$v/libsiammask_hip.so timeout 300 python -m pytest tests/test_gpu_seq.py -x -q -k "fused or uneven or repeated" 2>&1 | grep -E "passed|failed" | tail -2 | tee -a $O/pytest.txt
done
B="python3 bench.py --steps 300 --warmup 20 --no-also --no-cpu-baseline --no-long"
for rep in 1 2 3; do
  for arm in product reslate1 reslate2; do
    unset SMK_LIB; [ $arm != product ] && export SMK_LIB=$R/build_variants/$arm/libsiammask_hip.so
    timeout 120 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm', d['value'], d['ms_per_step'], 'seq', d['roofline']['avg_launch_us'])" | tee -a $O/ab.txt
  done
done
unset SMK_LIB
for arm in product reslate1 reslate2; do
  [ $arm != product ] && export SMK_LIB=$R/build_variants/$arm/libsiammask_hip.so
  echo "== $arm" >> $O/stamps.txt
  SMK_GRAPH=0 SMK_SEQ_CLK=2 timeout 120 python tools/measure/gpu_seqclk.py 2>&1 | grep "seq arrive\|seq clk2\|total" | grep "l3.2\|total" | tail -5 >> $O/stamps.txt
  unset SMK_LIB
done
cat $O/stamps.txt | cut -c1-330
