#!/bin/bash
# the driver's bench command once more behind the corrected PMC summary (roofline.traffic) -> gpurun_out/r05final/bench_driver_cmd.json
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05final; mkdir -p $O
export SMK_GRAPH=1
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --profile-out $O/layers_b8.json > $O/bench_driver_cmd.json 2> $O/bench.err; echo "driver-cmd bench exit $?"
python - <<PY
import json
d = json.loads(open("$O/bench_driver_cmd.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("driver cmd:", d["value"], d["ms_per_step"], d.get("value_200_steps"), "serial:", d.get("serial_steps"), "latency:", d.get("latency"))
print("roofline:", r["kernel"], r["frac"], r["avg_launch_us"], "traffic", r["traffic"], r.get("traffic_over_external"), "rocprofv3", r.get("rocprofv3"))
print("also:", {k: (v.get("fps"), v.get("ms_per_step"), v.get("mfma_frac"), v.get("pipelined")) for k, v in (d.get("also") or {}).items()})
PY
