#!/bin/bash
# Round 6: B = 64 step with conv_pp by the rule (pp=1), off (pp=0), and on every eligible layer incl. the Bottlenecks' 1x1 convolutions instead of
# the fused pair launches (pp=2,pair_launch=0); interleaved twice
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06f; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -k "b64" 2>&1 | tail -3 | tee $O/pytest.txt
for t in pp=0 pp=1 pp=2,pair_launch=0 pp=0 pp=1 pp=2,pair_launch=0; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-long --workload sharp_b64_f16 --tune $t > $O/b64_$t.json 2>> $O/bench.err
  python - <<PY
import json
d = json.loads(open("$O/b64_$t.json").read().strip().splitlines()[-1])
print("$t", d["value"], d["ms_per_step"], [(k["kernel"], k["launches"], round(k["us_per_step"], 1), round(k["achieved"], 1)) for k in d["roofline"]["kernels"][:7]])
PY
done 2>&1 | tee $O/b64_ab.txt
tail -3 $O/bench.err
