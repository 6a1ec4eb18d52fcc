#!/bin/bash
# Round 6: per-layer table of the split-operand context at B = 8
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06l; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also --no-long --workload sharp_b8_f16x3 --profile-out $O/layers_b8_f16x3.json > $O/bench.json 2>> $O/bench.err
python - <<PY
import json
d = json.load(open("$O/layers_b8_f16x3.json"))
tot = 0
for l in d["layers"]:
    us = l["ms"] * 1000 / l["calls"]; tot += us
    print("%-28s %-44s %8.1f us  %7.1f TF(alg)  %7.1f GB/s" % (l["id"][:28], l["kernel"][:44], us, l["flop"] / l["calls"] / us / 1e6 if us else 0, l["bytes"] / l["calls"] / us / 1e3 if us else 0))
print("sum %.1f us" % tot)
PY
tail -2 $O/bench.err
