#!/bin/bash
# Round 6: wave priority (s_setprio) of a pipelined step's main-part launches (stem_pool, l1_block, per-layer convolutions): they share SIMDs with the previous frame's tail
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06bb; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
l = d.get("latency") or {}
print("%-30s %9.1f frames/s  %.4f ms   latency box %s mask %s" % (sys.argv[2], d["value"], d["ms_per_step"], l.get("box_ms_median"), l.get("mask_ms_median")))
PY
}
for wl in sharp_b8_f16 sharp_b1_f16 sharp_b64_f16 sharp_b8_f16x3; do
  for t in 0 3 1 0 3 1; do
    timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --workload $wl --no-cpu-baseline --no-also --no-long --tune main_prio=$t > $O/${wl}_$t.json 2>> $O/bench.err
    line $O/${wl}_$t.json "$wl main_prio=$t"
  done
done 2>&1 | tee $O/ab.txt
tail -2 $O/bench.err
