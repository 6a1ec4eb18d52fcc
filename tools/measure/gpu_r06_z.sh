#!/bin/bash
# Round 6: queue priority of the pipelined step's side stream (the tail shares the chip with the next frame's front end: 122 us where the halves are 69 and 97 alone)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06z; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
l = d.get("latency") or {}
print("%-30s %9.1f frames/s  %.4f ms   latency box %s mask %s" % (sys.argv[2], d["value"], d["ms_per_step"], l.get("box_ms_median"), l.get("mask_ms_median")))
PY
}
for wl in sharp_b8_f16 sharp_b1_f16 sharp_b64_f16 sharp_b8_f16x3; do
  for t in 0 1 2 0 1 2; do
    timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --workload $wl --no-cpu-baseline --no-also --no-long --tune pipe_prio=$t > $O/${wl}_$t.json 2>> $O/bench.err
    line $O/${wl}_$t.json "$wl pipe_prio=$t"
  done
done 2>&1 | tee $O/ab.txt
tail -2 $O/bench.err
