#!/bin/bash
# A/B sweep of library tuning knobs: one bench run (with per-layer profile) per setting.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/sweep
python -m pytest tests -m gpu -q --tb=short -x -k "ops or fp32_matches" > gpurun_out/sweep/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/sweep/pytest.log
tail -3 gpurun_out/sweep/pytest.log
i=0
for T in "$@"; do
  i=$((i+1))
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-also --tune "$T" ${WL:+--workload $WL} --profile-out gpurun_out/sweep/layers_$i.json > gpurun_out/sweep/bench_$i.json 2> gpurun_out/sweep/err_$i.log
  echo "$T" > gpurun_out/sweep/tune_$i.txt
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/sweep/bench_$i.json"))
    print("$T", d["value"], "fps", d["ms_per_step"], "ms/step  conv TF", d["roofline"]["achieved"], "kernel ms", d["roofline"]["kernel_ms_per_step"])
except Exception as e:
    print("$T FAILED", e)
PY
done
