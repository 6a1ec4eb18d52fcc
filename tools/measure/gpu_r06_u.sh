#!/bin/bash
# Round 6: the tail gate's clock armed by the main gate: pipeline suite + the headline line (the gate kernels changed)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06u; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ring.py tests/test_gpu_x3.py -x -q 2>&1 | tail -5 | tee $O/pytest.txt
for i in 1 2; do
timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-also --no-long > $O/b8_$i.json 2>> $O/bench.err
python - <<PY
import json
d = json.loads(open("$O/b8_$i.json").read().strip().splitlines()[-1])
print("fp16 B=8", d["value"], d["ms_per_step"], (d.get("serial_steps") or {}).get("ms_per_step"))
PY
done 2>&1 | tee $O/b8.txt
