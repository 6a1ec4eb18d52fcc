#!/bin/bash
# Round 6: split tensors stored as two planes [hi | lo] (the gather reads hi twice) + the row-segment dw-xcorr: parity, then the f16x3 bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06p; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 1500 python -m pytest tests/test_gpu_x3.py tests/test_gpu_ops.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -6 | tee $O/pytest.txt
for i in 1 2; do
timeout 300 python bench.py --gpus 1 --steps 100 --warmup 10 --workload sharp_b8_f16x3 --no-cpu-baseline --no-also --no-long > $O/b8_x3_$i.json 2>> $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/b8_x3_$i.json").read().strip().splitlines()[-1])
print("f16x3", d["value"], d["ms_per_step"], d.get("serial_steps", {}).get("ms_per_step"))
for k in d["roofline"]["kernels"]: print("   %-48s %2d %8.1f us" % (k["kernel"], k["launches"], k["us_per_step"]))
PY
done 2>&1 | tee $O/b8_x3.txt
timeout 600 python -m pytest tests/test_gpu_tools.py -x -q -k "f16x3" 2>&1 | tail -3 | tee -a $O/pytest.txt
tail -3 $O/bench.err
