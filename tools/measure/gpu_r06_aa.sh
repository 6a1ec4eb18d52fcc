#!/bin/bash
# Round 6: split-operand packs in FUSED order for conv_wreg_kernel (a hi activation tile staged once for its two products): parity, then A/B x3_fused 0 / 1
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06aa; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 1500 python -m pytest tests/test_gpu_x3.py -x -q -s 2>&1 | grep -E "x3 conv|f16x3|passed|failed|Error|assert" | tail -30 | tee $O/pytest.txt
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ks = {k["kernel"].replace("conv_wreg<f16,", "").rstrip(">"): round(k["us_per_step"], 1) for k in d["roofline"]["kernels"] if "wreg" in k["kernel"]}
print("%-30s %9.1f frames/s  %.4f ms  %s" % (sys.argv[2], d["value"], d["ms_per_step"], ks))
PY
}
for t in x3_fused=0 x3_fused=1 x3_fused=0 x3_fused=1; do
  timeout 300 python bench.py --gpus 1 --steps 100 --warmup 10 --workload sharp_b8_f16x3 --no-cpu-baseline --no-also --no-long --tune $t > $O/x3_$t.json 2>> $O/bench.err
  line $O/x3_$t.json "sharp_b8_f16x3 $t"
done 2>&1 | tee $O/ab.txt
timeout 600 python -m pytest tests/test_gpu_tools.py -x -q -k "f16x3" 2>&1 | tail -2 | tee -a $O/pytest.txt
tail -2 $O/bench.err
