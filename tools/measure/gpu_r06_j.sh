#!/bin/bash
# Round 6: first contact of the split-operand fp16 context (dtype f16x3): unit convolutions, B = 2 / 8 end to end vs the fp64 oracle, the 1024-stream
# argmax statistic, and its step time beside the fp32 context
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06j; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 900 python -m pytest tests/test_gpu_x3.py -x -q -s -k "convolution" 2>&1 | grep -E "x3 conv|passed|failed|Error|error|assert" | tail -14 | tee $O/pytest_conv.txt
timeout 900 python -m pytest tests/test_gpu_x3.py -x -q -s -k "end_to_end" 2>&1 | grep -E "f16x3 end|passed|failed|Error|error|assert|over the" | tail -14 | tee $O/pytest_e2e.txt
timeout 1500 python -m pytest tests/test_gpu_x3.py -x -q -s -k "1024" 2>&1 | grep -E "f16x3 vs|passed|failed|Error|error|assert" | tail -8 | tee $O/pytest_argmax.txt
for wl in sharp_b8_f16x3 sharp_b8_f32; do
  timeout 300 python bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline --no-also --no-long --workload $wl > $O/bench_$wl.json 2>> $O/bench.err
  python - <<PY
import json
d = json.loads(open("$O/bench_$wl.json").read().strip().splitlines()[-1])
print("$wl", d["value"], d["ms_per_step"], [(k["kernel"], k["launches"], round(k["us_per_step"], 1), round(k["achieved"], 1)) for k in d["roofline"]["kernels"][:8]])
PY
done 2>&1 | tee $O/bench.txt
tail -5 $O/bench.err
