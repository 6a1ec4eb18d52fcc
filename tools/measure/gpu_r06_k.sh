#!/bin/bash
# Round 6: split-operand context with conv_wreg_kernel allowed for its convolutions: parity (unit, end to end, 1024 streams), then wreg=0 / wreg=1 step times
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06k; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 1800 python -m pytest tests/test_gpu_x3.py -x -q -s 2>&1 | grep -E "x3 conv|f16x3|passed|failed|Error|error|assert" | tail -20 | tee $O/pytest.txt
for t in wreg=0 wreg=1 wreg=0 wreg=1; do
  timeout 300 python bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline --no-also --no-long --workload sharp_b8_f16x3 --tune $t > $O/bench_$t.json 2>> $O/bench.err
  python - <<PY
import json
d = json.loads(open("$O/bench_$t.json").read().strip().splitlines()[-1])
print("$t", d["value"], d["ms_per_step"], [(k["kernel"], k["launches"], round(k["us_per_step"], 1), round(k["achieved"], 1)) for k in d["roofline"]["kernels"][:7]])
PY
done 2>&1 | tee $O/bench.txt
tail -3 $O/bench.err
