#!/bin/bash
# Round 6: full GPU suite on the current tree; B = 1 / 2 with the persistent sequence (one image per XCD: seq_min_batch=1) against the per-layer launches
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06x; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest.txt
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
ks = {k["kernel"]: round(k["us_per_step"], 1) for k in d["roofline"]["kernels"] if "seq" in k["kernel"]}
print("%-34s %9.1f frames/s  %.4f ms  %s" % (sys.argv[2], d["value"], d["ms_per_step"], ks))
PY
}
for t in seq_min_batch=5 seq_min_batch=1 seq_min_batch=5 seq_min_batch=1; do
  timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --workload sharp_b1_f16 --no-cpu-baseline --no-also --no-long --tune $t > $O/b1_$t.json 2>> $O/bench.err
  line $O/b1_$t.json "sharp_b1_f16 $t"
done 2>&1 | tee $O/b1_seq.txt
tail -2 $O/bench.err
