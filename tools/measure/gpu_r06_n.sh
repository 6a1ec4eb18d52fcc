#!/bin/bash
# Round 6: decode_kernel with wave-butterfly reduction + one-round-trip last arriver: parity (device decode vs the unchanged tool's fixtures, rings, pipelined steps),
# the f16x3 tool trajectory, then the driver command (decode row)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06n; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 1800 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_ring.py tests/test_gpu_pipeline.py tests/test_gpu_e2e.py tests/test_gpu_tracker.py tests/test_gpu_argmax.py -x -q 2>&1 | tail -4 | tee $O/pytest.txt
timeout 1500 python -m pytest tests/test_gpu_tools.py -x -q 2>&1 | tail -4 | tee $O/pytest_tools.txt
cat gpurun_out/tools_on_mi355x_f16x3.json 2>/dev/null | head -30 | tee $O/tools_f16x3.json
for i in 1 2; do
timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-also --no-long > $O/b8_$i.json 2>> $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/b8_$i.json").read().strip().splitlines()[-1])
ks = {k["kernel"]: round(k["us_per_step"], 1) for k in d["roofline"]["kernels"]}
print(d["value"], d["ms_per_step"], ks)
PY
done 2>&1 | tee $O/b8.txt
timeout 300 python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline --no-also --no-long --workload sharp_b1_f16 > $O/b1.json 2>> $O/bench.err
python -c "
import json; d = json.loads(open('$O/b1.json').read().strip().splitlines()[-1]); print('b1', d['value'], d['ms_per_step'], {k['kernel']: round(k['us_per_step'], 1) for k in d['roofline']['kernels']})" | tee -a $O/b8.txt
tail -3 $O/bench.err
