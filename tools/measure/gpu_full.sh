#!/bin/bash
# full GPU test-suite + bench variants
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/full
python -m pytest tests -m gpu -q --tb=short > gpurun_out/full/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/full/pytest.log
tail -15 gpurun_out/full/pytest.log
i=0
for ARGS in "$@"; do
  i=$((i+1))
  python bench.py --no-cpu-baseline --no-also $ARGS --profile-out gpurun_out/full/layers_$i.json > gpurun_out/full/bench_$i.json 2> gpurun_out/full/err_$i.log
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/full/bench_$i.json"))
    print("$ARGS ->", d["value"], "fps", d["ms_per_step"], "ms/step  conv TF", d["roofline"]["achieved"], "kernel ms", d["roofline"]["kernel_ms_per_step"])
except Exception as e:
    print("$ARGS FAILED", e); print(open("gpurun_out/full/err_$i.log").read()[-1500:])
PY
done
