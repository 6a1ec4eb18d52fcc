#!/bin/bash
# Round 5: mask_head_kernel for B > 16: parity + A/B at B = 64 (+ its per-launch time)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05l; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 900 python -m pytest tests/test_gpu_mask_head.py -x -q -s 2>&1 | grep -E "mask_head vs|passed|failed|Error|error" | tail -12 | tee $O/pytest.txt
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -k "b64" 2>&1 | tail -3 | tee -a $O/pytest.txt
for t in mask_kernel=1 mask_kernel=0 mask_kernel=1 mask_kernel=0; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-long --workload sharp_b64_f16 --tune $t > $O/b64_$t.json 2>> $O/bench.err
  python - <<PY
import json
d = json.loads(open("$O/b64_$t.json").read().strip().splitlines()[-1])
print("$t", d["value"], d["ms_per_step"], [ (k["kernel"], k["us_per_step"], k["achieved"]) for k in d["roofline"]["kernels"] if "mask" in k["kernel"] or "nchw" in k["kernel"]])
PY
done
tail -3 $O/bench.err
