#!/bin/bash
# Round 4: B = 64 -- do the merged launches (shortcut + conv1, Refine's window convolutions, v*.2, cls3 + loc3) still pay when every
# member fills the chip by itself?  bench lines with merge on / off, per-kernel tables.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04j; mkdir -p $O
export SMK_GRAPH=1
for t in "" "merge=0" "" "merge=0"; do
  X=""; [ -n "$t" ] && X="--tune $t"
  timeout 150 python3 bench.py --workload sharp_b64_f16 --steps 30 --warmup 5 --prewarm-seconds 1 --no-also --no-cpu-baseline --no-long $X 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('tune [%s]' % '$t', d['value'], d['ms_per_step'])
for r in k[:14]: print('     %-52s %5.1f launches %8.1f us' % (r['kernel'][:52], r['launches'], r['us_per_step']))" | tee -a $O/b64_merge.txt
done
