#!/bin/bash
# Round 6: does the pipelined B = 8 step's tail run BESIDE the next frame's front end when the front end's kernels are limited to one workgroup
# per CU (smk_tune front_occ1: bit 0 l1_block, bit 1 stem_pool)?  off / 1 / 3 / off / 1 / 3, driver command
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06g; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
for t in front_occ1=0 front_occ1=1 front_occ1=3 front_occ1=0 front_occ1=1 front_occ1=3; do
  timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-also --no-long --tune $t > $O/b8_$t.json 2>> $O/bench.err
  python - <<PY
import json
d = json.loads(open("$O/b8_$t.json").read().strip().splitlines()[-1])
ks = {k["kernel"]: round(k["us_per_step"], 1) for k in d["roofline"]["kernels"]}
print("$t", d["value"], d["ms_per_step"], "serial", d.get("serial_steps", {}).get("ms_per_step"), "lat", d.get("latency", {}).get("box_ms_median"), d.get("latency", {}).get("mask_ms_median"), {k: v for k, v in ks.items() if "l1" in k or "stem" in k or "chain" in k})
PY
done 2>&1 | tee $O/b8_ab.txt
tail -3 $O/bench.err
