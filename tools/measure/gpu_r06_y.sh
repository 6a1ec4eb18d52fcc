#!/bin/bash
# Round 6: Refine's two merged front launches on 32 x 64 tiles (rf_wreg: 3 = both on 64 x 64; +128 = the window convolutions + deconv on 32 x 64; +2048 = the v*.2 launch)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06y; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ks = {k["kernel"].replace("conv_wreg<f16,", "").rstrip(">"): round(k["us_per_step"], 1) for k in d["roofline"]["kernels"] if "merged" in k["kernel"]}
print("%-30s %9.1f frames/s  %.4f ms  %s" % (sys.argv[2], d["value"], d["ms_per_step"], ks))
PY
}
for wl in sharp_b8_f16 sharp_b1_f16; do
  for t in 3 131 2051 2179 3 131 2051 2179; do
    timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --workload $wl --no-cpu-baseline --no-also --no-long --tune rf_wreg=$t > $O/${wl}_$t.json 2>> $O/bench.err
    line $O/${wl}_$t.json "$wl rf_wreg=$t"
  done
done 2>&1 | tee $O/ab.txt
tail -2 $O/bench.err
