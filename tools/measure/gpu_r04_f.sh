#!/bin/bash
# Round 4, sixth GPU contact: B = 64 -- every register-fed layer forced onto one workgroup shape (smk_tune wreg = code + 1:
# 64x256, 64x128, 64x64, 128x256, 128x128, 128x64) against the per-layer rule: is one workgroup per CU (128-row tiles, 139 KB of
# LDS) what holds the HBM-bound 1x1 layers at 2.2 TB/s?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04f; mkdir -p $O
export SMK_GRAPH=1
for t in "" "wreg=2" "wreg=3" "wreg=5" "wreg=6" "wreg=0" ""; do
  X=""; [ -n "$t" ] && X="--tune $t"
  timeout 150 python3 bench.py --workload sharp_b64_f16 --steps 30 --warmup 5 --prewarm-seconds 1 --no-also --no-cpu-baseline --no-long $X 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('tune [%s]' % '$t', d['value'], d['ms_per_step'], ' | '.join('%s %.0f' % (r['kernel'][:28], r['us_per_step']) for r in k[:5]))" | tee -a $O/b64_tiles.txt
done
