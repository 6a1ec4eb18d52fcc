#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06t; rm -rf $O; mkdir -p $O
export SMK_LIB=$R/build_variants/measure/siammask_amd/libsiammask_hip.so SMK_GRAPH=1
timeout 600 python tools/measure/gpu_wreg_ring_depth.py 2>&1 | tee $O/ring_depth_layers.txt
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-30s %9.1f frames/s  %.4f ms" % (sys.argv[2], d["value"], d["ms_per_step"]))
PY
}
for wl in sharp_b8_f16x3 sharp_b1_f16; do
  for t in 3 4 5 6 7 3 4 5 6 7; do
    timeout 300 python bench.py --gpus 1 --steps 100 --warmup 10 --workload $wl --no-cpu-baseline --no-also --no-long --tune wreg_stages=$t > $O/${wl}_$t.json 2>> $O/bench.err
    line $O/${wl}_$t.json "$wl wreg_stages=$t"
  done
done 2>&1 | tee $O/ring_depth_steps.txt
tail -2 $O/bench.err
