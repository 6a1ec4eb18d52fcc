#!/bin/bash
# Round 4, fifth GPU contact: result ring folded into the decode / chain launches (no commit launch), the tile of the merged
# v*.2 launch (rf_tile2), then the ring / no-ring A/B of bench.py again.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04e; mkdir -p $O
export SMK_GRAPH=1
timeout 600 python -m pytest tests/test_gpu_ring.py tests/test_gpu_dist.py "tests/test_gpu_e2e.py::test_refine_chain_equals_layer_path" tests/test_gpu_dropin.py -x -q 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/pytest_ring.txt
timeout 300 python tools/measure/gpu_knob_ab.py rf_tile2 8 0,2 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/ab_rf_tile2.txt
timeout 300 python tools/measure/gpu_knob_ab.py rf_tile2 8 0,1 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O/ab_rf_tile2.txt
B="python3 bench.py --steps 300 --warmup 20 --no-also --no-cpu-baseline --no-long"
for rep in 1 2 3; do
  for arm in ring noring; do
    X=""; [ $arm = noring ] && X="--no-ring"
    timeout 120 $B $X 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm', d['value'], d['ms_per_step'], 'seq', d['roofline']['avg_launch_us'], 'launches', d['roofline']['launches_per_step_all_kernels'])" | tee -a $O/ab_ring.txt
  done
done
