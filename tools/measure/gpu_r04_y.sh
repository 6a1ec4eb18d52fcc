#!/bin/bash
# Round 4: the triple routine now lives in its own instantiation of conv_seq_kernel (T3 = 1); the product instantiation against the
# variant compiled without any of it (SMK_NO_TRIPLES no longer exists: build_variants/notriple was built from the sources before), and the tests.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04y; mkdir -p $O
export SMK_GRAPH=1
timeout 900 python -m pytest tests/test_gpu_seq.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_seq.txt
B="python3 bench.py --steps 400 --warmup 20 --no-also --no-cpu-baseline --no-long"
for rep in 1 2 3; do
  for arm in product notriple; do
    unset SMK_LIB; [ $arm != product ] && export SMK_LIB=$R/build_variants/$arm/libsiammask_hip.so
    timeout 120 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$arm', d['value'], d['ms_per_step'], 'seq', d['roofline']['avg_launch_us'])" | tee -a $O/ab.txt
  done
done
unset SMK_LIB
timeout 300 python tools/measure/gpu_knob_ab.py seq_fuse3 8 0,1 2>&1 | grep "ms/step" | tee $O/knob_ab.txt
