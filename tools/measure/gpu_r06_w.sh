#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06w; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "wreg" 2>&1 | tail -3 | tee $O/pytest.txt
timeout 600 python tools/measure/gpu_wreg32.py 2>&1 | tee $O/wreg32_layers.txt
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-34s %9.1f frames/s  %.4f ms" % (sys.argv[2], d["value"], d["ms_per_step"]))
PY
}
for wl in sharp_b1_f16 sharp_b8_f16; do
  for t in wreg32=0 wreg32=100 wreg32=140 wreg32=260 wreg32=0 wreg32=100 wreg32=140 wreg32=260; do
    timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --workload $wl --no-cpu-baseline --no-also --no-long --tune $t > $O/${wl}_$t.json 2>> $O/bench.err
    line $O/${wl}_$t.json "$wl $t"
  done
done 2>&1 | tee $O/wreg32_steps.txt
tail -2 $O/bench.err
