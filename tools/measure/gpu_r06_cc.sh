#!/bin/bash
# (RECORD ONLY: the smk_tune tail_split knob and its code were removed again after this measurement -- profiles/r06cc_mask_head_outside_the_chain_launch.txt)
# Round 6: pipelined step with the mask head as its own launch on a second side stream (own gate + completion semaphore; 64 KB of LDS per workgroup) instead of inside chain_mask_kernel (140 KB for each of its ~650
# workgroups: none of them shares a CU with the next frame's front end): parity of the pipeline suite under the knob, then A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06cc; rm -rf $O; mkdir -p $O
export SMK_GRAPH=1
SMK_TUNE=tail_split=1 timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ring.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
l = d.get("latency") or {}
print("%-34s %9.1f frames/s  %.4f ms   latency box %s mask %s" % (sys.argv[2], d["value"], d["ms_per_step"], l.get("box_ms_median"), l.get("mask_ms_median")))
PY
}
for wl in sharp_b8_f16 sharp_b1_f16 sharp_b8_f16x3 sharp_b16_f16; do
  for t in 0 1 0 1; do
    timeout 300 python bench.py --gpus 1 --steps 200 --warmup 20 --workload $wl --no-cpu-baseline --no-also --no-long --tune tail_split=$t > $O/${wl}_$t.json 2>> $O/bench.err
    line $O/${wl}_$t.json "$wl tail_split=$t"
  done
done 2>&1 | tee $O/ab.txt
tail -2 $O/bench.err
