#!/bin/bash
# Round 4, first GPU contact: the full GPU suite at HEAD (engine changes: same-call sequence failure check, f32 arena aliases,
# pair-fusion rule), the new gates (fp16 argmax agreement, side-stream gather overlap, injected sequence failure), the
# FETCH_SIZE / WRITE_SIZE calibration, the default bench line (vendor baseline, argmax agreement, per-kernel table) and the
# per-phase stamps of the sequence on this box.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04a; mkdir -p $O
export SMK_GRAPH=1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -15 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
bash tools/measure/gpu_pmc_calib.sh 2>&1 | tail -4
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --profile-out $O/layers_b8.json > $O/bench_driver_cmd.json 2> $O/bench.err; echo "driver-cmd bench exit $?"
tail -3 $O/bench.err
SMK_GRAPH=0 SMK_SEQ_CLK=2 timeout 120 python tools/measure/gpu_seqclk.py > $O/seqclk.txt 2>&1
python - <<PY
import json
d=json.loads(open("$O/bench_driver_cmd.json").read().strip().splitlines()[-1])
print("driver cmd:", d["value"], d["ms_per_step"], d.get("value_200_steps"), d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
print("also:", {k: (v.get("fps"), v.get("ms_per_step"), v.get("mfma_frac")) for k, v in (d.get("also") or {}).items()})
print("vendor:", d.get("vendor_baseline"))
print("argmax:", d.get("argmax_agreement"))
print("cpu:", (d.get("cpu_baseline") or {}).get("value"))
for r in d["roofline"].get("kernels", []): print("  ", r)
PY
