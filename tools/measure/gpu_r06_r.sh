#!/bin/bash
# Round 6: conv_wreg with both operand streams eight k-steps ahead on every tile shape (smk_tune wreg_stages=8): parity (bit-equal to the 3- / 4-deep rings),
# then A/B on the three regimes that launch the narrow tiles: B = 1 fp16, B = 8 fp16 (Refine's window convolutions), B = 8 f16x3
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06r; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "wreg" 2>&1 | tail -4 | tee $O/pytest.txt
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ks = {}
for k in d["roofline"]["kernels"]:
    if "wreg" in k["kernel"]: ks[k["kernel"].replace("conv_wreg<f16,", "").rstrip(">")] = round(k["us_per_step"], 1)
print("%-30s %9.1f frames/s  %.4f ms   %s" % (sys.argv[2], d["value"], d["ms_per_step"], ks))
PY
}
for wl in sharp_b1_f16 sharp_b8_f16 sharp_b8_f16x3; do
  for t in wreg_stages=0 wreg_stages=8 wreg_stages=0 wreg_stages=8; do
    timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --workload $wl --no-cpu-baseline --no-also --no-long --tune $t > $O/${wl}_$t.json 2>> $O/bench.err
    line $O/${wl}_$t.json "$wl $t"
  done
done 2>&1 | tee $O/ab.txt
tail -3 $O/bench.err
