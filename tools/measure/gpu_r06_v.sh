#!/bin/bash
# Round 6: the narrow tiles wait for their activation refills (profiles/r06t): the two other ways the rows can reach the ring, as whole-step A/Bs
# (a_stage=1: global -> VGPR -> ds_write instead of LDS-DMA; npw=2: two producer waves instead of four)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06v; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-30s %9.1f frames/s  %.4f ms" % (sys.argv[2], d["value"], d["ms_per_step"]))
PY
}
for wl in sharp_b8_f16x3 sharp_b1_f16 sharp_b8_f16; do
  for t in a_stage=0 a_stage=1 npw=2 a_stage=0 a_stage=1 npw=2; do
    timeout 300 python bench.py --gpus 1 --steps 100 --warmup 10 --workload $wl --no-cpu-baseline --no-also --no-long --tune $t > $O/${wl}_$t.json 2>> $O/bench.err
    line $O/${wl}_$t.json "$wl $t"
  done
done 2>&1 | tee $O/ab.txt
tail -2 $O/bench.err
