#!/bin/bash
# Round 5: the main gate in front of the heads for batches outside the persistent sequence (pipe_late): parity + A/B at B = 1 / 64 / 3 / 32
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r05h; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ring.py tests/test_gpu_tracker.py -x -q 2>&1 | tail -5 | tee $O/pytest.txt
for wl in sharp_b1_f16:50 sharp_b64_f16:20 sharp_b8_f32:20; do
  w=${wl%%:*}; k=${wl##*:}
  for t in pipe_late=1 pipe_late=0 pipe_late=1 pipe_late=0; do
    timeout 300 python bench.py --gpus 1 --steps $k --warmup 5 --no-cpu-baseline --no-also --no-long --workload $w --tune $t > $O/${w}_$t.json 2>> $O/bench.err
    python - <<PY
import json
d = json.loads(open("$O/${w}_$t.json").read().strip().splitlines()[-1])
print("$w $t", d["value"], d["ms_per_step"], "lat:", d.get("latency"), "serial:", (d.get("serial_steps") or {}).get("ms_per_step"))
PY
  done
done
tail -3 $O/bench.err
