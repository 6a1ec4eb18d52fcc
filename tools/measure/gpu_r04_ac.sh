#!/bin/bash
# Round 4: after the lean product instantiation (+ the stamps moved to the CLK build): sequence tests, the measurement-build test, a_stage e2e test, bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04ac; mkdir -p $O
export SMK_GRAPH=1
timeout 900 python -m pytest tests/test_gpu_seq.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_seq.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -k "a_stage or bench_configuration or which_batches" 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_e2e.txt
B="python3 bench.py --steps 400 --warmup 20 --no-also --no-cpu-baseline --no-long"
for rep in 1 2 3; do
    timeout 120 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('product', d['value'], d['ms_per_step'], 'seq', d['roofline']['avg_launch_us'])" | tee -a $O/ab.txt
done
SMK_GRAPH=0 SMK_SEQ_CLK=1 timeout 120 python tools/measure/gpu_seqclk.py 2>&1 | grep "total" | tail -3
